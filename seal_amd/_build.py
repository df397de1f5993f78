"""Builds libsealfm.so (hipcc, gfx950 only) in-tree next to its sources: one object per source file (recompiled only
when the sha256 of that file, the headers and the flags changed -- not mtimes --, in parallel), then one link; the library carries a stamp
with the digest of everything it was built from, and `stale()` compares that stamp with the sources present."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libsealfm.so")
SOURCES = ["fmi_host.cpp", "fmi_evidence.cpp", "fmi_agg_pack.cpp", "fmi_sdsl.cpp", "fmi_kernels.hip", "fmi_aggregate.hip", "fmi_build_gpu.hip",
           "bart_kernels.hip", "hgemm_kernels.hip", "fmi_upload.hip"]
HEADERS = ["fmi_internal.h", "fmi_device.h", "fmi_agg.h", os.path.join("..", "..", "include", "sealfm.h"),
           os.path.join("..", "..", "include", "sealnn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required; this package targets gfx950 only)")


STAMP = os.path.join(CSRC, "libsealfm.so.sha256")


def source_digest() -> str:
    """sha256 over every source, header and the flags: what libsealfm.so must have been built from.  (mtimes say nothing about a
    git-ignored binary that travels with a snapshot: a stale library with a fresh mtime would pass an mtime check.)"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        h.update(f.encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_digest() -> str:
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return ""


def stale() -> bool:
    return not os.path.exists(LIB) or built_digest() != source_digest()


def _file_digest(path: str) -> str:
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in [path] + [os.path.join(CSRC, x) for x in HEADERS]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        if verbose:
            print("libsealfm.so is up to date with its sources (sha256 %s)" % built_digest()[:16], flush=True)
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    digest = source_digest()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, src + ".o")
        src_path = os.path.join(CSRC, src)
        want = _file_digest(src_path)
        try:
            with open(obj + ".sha256") as f:
                have = f.read().strip()
        except OSError:
            have = ""
        if force or not os.path.exists(obj) or have != want:
            cmd = [hipcc] + FLAGS + ["-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(obj + ".sha256", "w") as f:
                f.write(want)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    if os.path.exists(STAMP):
        os.remove(STAMP)                 # never a stamp beside a library it does not describe
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
