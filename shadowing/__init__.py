"""Drop-in alias: `from shadowing import PathShadowing, Foveal, RelativeMSE, ...` as
the reference's notebooks and README do, served by shadowing_amd."""
from shadowing_amd import *  # noqa: F401,F403
from shadowing_amd import __all__, __version__  # noqa: F401
