"""In-tree build of liburnn_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "liburnn_hip.so")
SOURCES = ["urnn_gemm.hip", "urnn_small.hip", "urnn_elem.hip", "urnn_train.hip", "urnn_api.hip"]
HEADERS = ["urnn_common.h", "urnn_kernels.h", os.path.join("..", "..", "include", "urnn_hip.h")]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False, extra_flags=(), out=None, tag=""):
    """Compile the HIP sources to ``u-rnn_amd/liburnn_hip.so``; returns the path.  ``extra_flags`` / ``out`` /
    ``tag`` build tuning variants next to the product library (development only)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    LIB = out or globals()["LIB"]
    if not force and os.path.isfile(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s).replace(".hip", tag + ".o"))
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra_flags, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
