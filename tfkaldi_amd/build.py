"""Build the gfx950 HIP engine (libtfkaldi_hip.so) in-tree with hipcc.

The library is the C-ABI drop-in boundary declared in include/tfkaldi_hip.h.  hipcc cross-compiles
for gfx950 without a GPU, so this runs in the build container and the .so travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libtfkaldi_hip.so"
SOURCES = ["gemm_f32.hip", "gemm_bf16.hip", "kernels.hip", "ctc.hip", "engine.hip", "features.hip"]
HEADERS = ["gemm_f32.h", "gemm_bf16.h", "kernels.h", "ctc.h", os.path.join("..", "..", "include", "tfkaldi_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


TRAFFIC_STAMP_SOURCES = ("gemm_f32.hip", "gemm_f32.h")


def csrc_hash():
    """sha256 (first 16 hex digits) over the sources the fp32 GEMM kernels compile from -- nothing else.  It stamps
    profiles/hbm_traffic.json, the PMC traffic figure of bench.py's dominant kernel (gemm_f32_dual / gemm_f32_kernel):
    the figure is valid exactly as long as those kernels are unchanged.  (Round 2 hashed every source plus the public
    header, so an enum added for the feature path invalidated the GEMM's traffic record; tests/test_host_logic.py now
    fails when the committed record and this hash disagree.)"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(TRAFFIC_STAMP_SOURCES):
        with open(os.path.join(CSRC, name), "rb") as fid:
            h.update(name.encode() + b"\0" + fid.read())
    return h.hexdigest()[:16]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 engine")
    return exe


def _stale():
    out = lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force=False, verbose=False):
    """Compile csrc/*.hip -> lib/libtfkaldi_hip.so (skipped when up to date). Returns the path."""
    if not force and not _stale():
        return lib_path()
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = lib_path() + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    os.replace(tmp, lib_path())
    return lib_path()


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
