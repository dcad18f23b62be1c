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
SOURCES = ["gemm_f32.hip", "gemm_bf16.hip", "kernels.hip", "ctc.hip", "engine.hip", "features.hip", "exchange.hip"]
HEADERS = ["gemm_f32.h", "gemm_bf16.h", "kernels.h", "ctc.h", "x3_layout.h", os.path.join("..", "..", "include", "tfkaldi_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


TRAFFIC_STAMP_SOURCES = ("gemm_f32.hip", "gemm_f32.h", "gemm_bf16.hip", "gemm_bf16.h", "x3_layout.h")


def csrc_hash():
    """sha256 (first 16 hex digits) over the sources the GEMM kernels compile from -- nothing else.  It stamps
    profiles/hbm_traffic.json, the PMC traffic figures of bench.py's dominant kernels (one record per configuration and
    arithmetic): a figure is valid exactly as long as those kernels are unchanged.  (Round 2 hashed every source plus the public
    header, so an enum added for the feature path invalidated the GEMM's traffic record; tests/test_host_logic.py fails when the
    committed record and this hash disagree.)"""
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


def _digest(names, extra=b""):
    import hashlib
    h = hashlib.sha256(extra)
    for name in names:
        with open(os.path.join(CSRC, name), "rb") as fid:
            h.update(os.path.basename(name).encode() + b"\0" + fid.read() + b"\0")
    return h.hexdigest()


def source_id():
    """Build provenance: sha256 (first 32 hex digits) over every source and header the library compiles from, the
    public header included, and the compiler flags.  hipcc bakes it into the library (TFK_BUILD_ID; tfk_build_id()
    returns it), _lib.load() refuses a library whose id is not this tree's, and build_native rebuilds on that --
    not on file times, which say nothing once the .so has travelled to another box."""
    return _digest(sorted(SOURCES) + sorted(HEADERS), " ".join(FLAGS).encode())[:32]


_ID_MARK = b"TFK_BUILD_ID="


def library_id(path=None):
    """the id baked into a built library, read from the file (no dlopen); None when there is none"""
    path = path or lib_path()
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fid:
        blob = fid.read()
    at = blob.find(_ID_MARK)
    if at < 0:
        return None
    return blob[at + len(_ID_MARK):at + len(_ID_MARK) + 32].decode("ascii", "replace")


def _stale():
    return library_id() != source_id()


def build_native(force=False, verbose=False):
    """Compile csrc/*.hip -> lib/libtfkaldi_hip.so (skipped when the library's build id is this tree's).  Objects are
    cached per translation unit under build/obj, keyed by the hash of the unit's source + every header + the flags.
    Returns the path."""
    if not force and not _stale():
        return lib_path()
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    build_id = source_id()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        flags = list(FLAGS)
        if src == "engine.hip":  # the unit that exports tfk_build_id()
            flags.append('-DTFK_BUILD_ID="%s"' % build_id)
        key = _digest([src] + sorted(HEADERS), " ".join(flags).encode())
        stamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
            return obj
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as fid:
            fid.write(key)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = lib_path() + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    os.replace(tmp, lib_path())
    if library_id() != build_id:
        raise RuntimeError("the linked library does not carry build id %s" % build_id)
    return lib_path()


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
