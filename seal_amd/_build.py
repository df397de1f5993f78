"""Builds libsealfm.so (hipcc, gfx950 only) in-tree next to its sources."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsealfm.so")
SOURCES = ["fmi_host.cpp", "fmi_evidence.cpp", "fmi_agg_pack.cpp", "fmi_sdsl.cpp", "fmi_kernels.hip", "fmi_aggregate.hip", "fmi_build_gpu.hip",
           "bart_kernels.hip"]
HEADERS = ["fmi_internal.h", "fmi_device.h", "fmi_agg.h", os.path.join("..", "..", "include", "sealfm.h"),
           os.path.join("..", "..", "include", "sealnn.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required; this package targets gfx950 only)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
