"""In-tree build of libpsh_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libpsh_hip.so"
INCLUDE = PKG.parent / "include"
SOURCES = [CSRC / "psh_scan.hip", CSRC / "psh_embed.hip", CSRC / "psh_select.hip", CSRC / "psh_capi.hip"]
DEPS = SOURCES + [CSRC / "psh_kernels.h", CSRC / "psh_device.h", INCLUDE / "psh.h"]

# -ffp-contract=off: nothing may be fused or re-associated that the source does not
# spell out -- bit-exact distances are what make the returned indices bit-exact.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def is_stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in DEPS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP kernels + C ABI into shadowing_amd/lib/libpsh_hip.so."""
    if not force and not is_stale():
        return LIB
    LIBDIR.mkdir(exist_ok=True)
    hipcc = hipcc_path()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)

    def compile_one(srcfile: Path) -> Path:
        obj = objdir / (srcfile.stem + ".o")
        cmd = [hipcc, *flags, f"-I{INCLUDE}", f"-I{CSRC}", "-c", str(srcfile), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {srcfile.name}:\n{res.stdout}\n{res.stderr}")
        return obj

    # the translation units are independent: compile them side by side, then link
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
