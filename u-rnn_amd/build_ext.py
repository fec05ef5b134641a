"""In-tree build of liburnn_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "liburnn_hip.so")
SOURCES = ["urnn_gemm.hip", "urnn_gemm_gates.hip", "urnn_gemm_cand.hip", "urnn_gemm_deconv.hip", "urnn_cand_fused.hip", "urnn_cand_gated.hip", "urnn_small.hip", "urnn_coop_tiles.hip",
           "urnn_tail.hip", "urnn_elem.hip", "urnn_train.hip", "urnn_api.hip"]
HEADERS = ["urnn_common.h", "urnn_kernels.h", "urnn_gemm.h", os.path.join("..", "..", "include", "urnn_hip.h")]


# No SLP vectorisation: on gfx950 it turns pairs of scalar fp32 operations into packed v_pk_mul_f32 / v_pk_fma_f32.  In the
# candidate GEMM (MFMAs, LDS-DMA, two waves per SIMD) those produced -- about once per 10^9 instructions, reproducibly on every box
# -- a wrong LOW element in lanes 16..31 of one wave: 16 pixel columns of one tile off by ~3e-3 in 10-90 of 6000 launches of the
# dec1 cell, whatever the activation code looked like (hardware transcendentals, padded inline asm, a VALU-only sigmoid) and with
# every ring slot poisoned with NaN until its DMA landed (no NaN ever appeared: the ring protocol was not the cause).  The same
# sources without packed instructions: 0 of 8000 launches, 0 of 30 whole-event rollouts, and 1.5 % faster (tools/diag_dec1.py,
# tools/stress_overlap.py; MI355X_MICROARCH.md prices the packed ops as an anti-lever beside MFMAs anyway).
NO_PACKED_F32 = ("-fno-slp-vectorize", "-DURNN_NO_PACKED_F32=1")   # urnn_common.h refuses to compile without the define


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False, extra_flags=(), out=None, tag=""):
    """Compile the HIP sources to ``u-rnn_amd/liburnn_hip.so``; returns the path.  ``extra_flags`` / ``out`` /
    ``tag`` build tuning variants next to the product library (development only)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    LIB = out or globals()["LIB"]
    if not force and not extra_flags and os.path.isfile(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    if extra_flags and (not out or not tag):
        raise ValueError("a build with extra flags needs its own library path (out=) and object tag (tag=): it must not replace the product library or its objects")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if any(f in ("-fslp-vectorize", "-fvectorize-slp") for f in extra_flags):
        raise ValueError("liburnn_hip must not be built with SLP vectorisation (packed fp32 next to MFMAs: DESIGN.md section 4.8)")
    objs = []
    procs = []
    hdrs = [d for d in deps if d not in srcs]
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s).replace(".hip", tag + ".o"))
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *NO_PACKED_F32, *extra_flags, "-c", s, "-o", o]
        # an object is reused only if it is newer than its sources AND was built by exactly this command line (recorded next to it):
        # an object left by a build with other flags must never be linked into the product library
        stamp = o + ".cmd"
        same_cmd = os.path.isfile(stamp) and open(stamp).read() == " ".join(cmd)
        if not force and same_cmd and os.path.isfile(o) and os.path.getmtime(o) >= _newest([s] + hdrs):
            continue                                  # this translation unit is up to date
        if verbose:
            print(" ".join(cmd))
        if os.path.isfile(stamp):
            os.remove(stamp)
        procs.append((subprocess.Popen(cmd), stamp, " ".join(cmd)))
    for p, stamp, line in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
        with open(stamp, "w") as f:
            f.write(line)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
