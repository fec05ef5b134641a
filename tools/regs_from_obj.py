#!/usr/bin/env python
"""Development aid: register / scratch report of every kernel in a built object (no recompilation):
   python tools/regs_from_obj.py u-rnn_amd/csrc/urnn_gemm.o [pattern]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj = os.path.abspath(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "."
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "o.o")
        os.symlink(obj, tmp)
        subprocess.run([LLVM + "llvm-objdump", "--offloading", tmp], capture_output=True, cwd=d)
        co = [f for f in os.listdir(d) if "amdgcn" in f][0]
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", os.path.join(d, co)], capture_output=True, text=True).stdout
    rx = re.compile(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                    r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", re.S)
    for ag, name, priv, sg, vg, sp in rx.findall(notes):
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("void ", "").replace("(ConvGemmParams)", "")
        if re.search(pat, dn):
            print(f"{dn:64s} vgpr {vg:>4} agpr {ag:>4} sgpr {sg:>4} scratch {priv:>5} spill {sp}")


if __name__ == "__main__":
    main()
