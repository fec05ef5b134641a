"""Build recipe for the CPU oracle (``gcc`` on oracle/urnn_oracle.c -> oracle/liburnn_oracle.so).

The reference is pure Python (no C/C++ sources), so there is nothing to compile
into ``oracle/_ref``; the oracle is pinned by goldens generated from the imported reference instead
(tests/golden/make_golden.py).  ``-march=x86-64-v3`` (AVX2+FMA) rather than ``native`` because the
.so built in the CPU container travels to the GPU box, whose host CPU differs.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_oracle(force=False):
    src = os.path.join(_HERE, "urnn_oracle.c")
    out = os.path.join(_HERE, "liburnn_oracle.so")
    if not force and os.path.isfile(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-o", out, src, "-lm"]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_oracle(force=True))
