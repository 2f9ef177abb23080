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
SOURCES = [CSRC / "psh_scan.hip", CSRC / "psh_fused.hip", CSRC / "psh_stream.hip", CSRC / "psh_lq.hip", CSRC / "psh_embed.hip", CSRC / "psh_embed_px.hip", CSRC / "psh_embed_mx.hip", CSRC / "psh_select.hip",
           CSRC / "psh_capi.hip", CSRC / "psh_comm.hip", CSRC / "psh_predict.hip", CSRC / "psh_prep.hip"]
DEPS = SOURCES + [CSRC / "psh_kernels.h", CSRC / "psh_device.h", INCLUDE / "psh.h"]

# -ffp-contract=off: nothing may be fused or re-associated that the source does not
# spell out -- bit-exact distances are what make the returned indices bit-exact.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
LINK_LIBS: list[str] = ["-ldl"]      # psh_comm.hip opens RCCL at run time (dlopen): no link-time dependency on it


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


TUNING_LIB = LIBDIR / "libpsh_hip_tuning.so"     # -DPSH_TUNING: geometry overrides + device time stamps for tools/


def source_hash() -> str:
    """SHA-256 over the sources the library is built from (file names + contents): what `is_stale` compares --
    modification times do not survive the copy to the GPU box, contents do."""
    import hashlib
    h = hashlib.sha256()
    for p in DEPS:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _stamp(lib: Path) -> Path:
    return lib.with_suffix(lib.suffix + ".srchash")


def is_stale(lib: Path | None = None) -> bool:
    """True when `lib` is missing or was built from other sources than the ones in the tree."""
    lib = lib or LIB
    stamp = _stamp(lib)
    if not lib.exists() or not stamp.exists():
        return True
    try:
        return stamp.read_text().strip() != source_hash()
    except OSError:
        # a lib-only deployment (no csrc/ or include/ beside the package): nothing to compare with -- the library is
        # taken as it is
        return False


def build(force: bool = False, verbose: bool = False, tuning: bool = False) -> Path:
    """Compile the HIP kernels + C ABI into shadowing_amd/lib/libpsh_hip.so (tuning=True: the instrumented
    libpsh_hip_tuning.so the scripts under tools/ load through PSH_LIB)."""
    lib = TUNING_LIB if tuning else LIB
    if not force and not is_stale(lib):
        return lib
    LIBDIR.mkdir(exist_ok=True)
    hipcc = hipcc_path()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DPSH_TUNING"] if tuning else [])
    objdir = LIBDIR / ("obj_tuning" if tuning else "obj")
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
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *map(str, objs), *LINK_LIBS, "-o", str(lib)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    _stamp(lib).write_text(source_hash() + "\n")
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, tuning="--tuning" in sys.argv))
