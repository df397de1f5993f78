"""Builds libsealfm.so (hipcc, gfx950 only) in-tree next to its sources: one object per source file (recompiled only
when that file or a header changed, in parallel), then one link."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libsealfm.so")
SOURCES = ["fmi_host.cpp", "fmi_evidence.cpp", "fmi_agg_pack.cpp", "fmi_sdsl.cpp", "fmi_kernels.hip", "fmi_aggregate.hip", "fmi_build_gpu.hip",
           "bart_kernels.hip"]
HEADERS = ["fmi_internal.h", "fmi_device.h", "fmi_agg.h", os.path.join("..", "..", "include", "sealfm.h"),
           os.path.join("..", "..", "include", "sealnn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required; this package targets gfx950 only)")


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc, hdr_t = _hipcc(), _newest_header()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, src + ".o")
        src_path = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src_path), hdr_t):
            cmd = [hipcc] + FLAGS + ["-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
