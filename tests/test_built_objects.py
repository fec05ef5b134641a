"""Properties of the BUILT gfx950 objects (CPU tests: they disassemble u-rnn_amd/csrc/*.o with llvm-objdump; build first).

1. DESIGN.md section 4.8: packed fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) next to MFMA results gave, about once per 1e9
   instructions, a wrong low element in 16 lanes of a wave.  Every unit is built with -fno-slp-vectorize; until round 6 only the
   bit-stability tests guarded that.  Here: no kernel that issues an MFMA contains a packed fp32 instruction.
2. Round 6 (profiles/r06_trace_coop_tiles.txt history): a cooperative kernel whose register allocation spilled ~150 registers to scratch
   ran 3-4x slower in EVERY phase (182 / 288 us against 62 / 73 us for the same work).  The cooperative kernels must not use scratch."""
import os
import re
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "u-rnn_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin/"
UNITS = ["urnn_gemm", "urnn_gemm_gates", "urnn_gemm_cand", "urnn_gemm_deconv", "urnn_cand_fused", "urnn_cand_gated", "urnn_small", "urnn_coop_tiles",
         "urnn_tail", "urnn_elem", "urnn_train", "urnn_api"]


def _device_code(unit):
    obj = os.path.join(CSRC, unit + ".o")
    if not os.path.isfile(obj) or not os.path.isfile(LLVM + "llvm-objdump"):
        pytest.skip(f"{obj} not built (python -c 'import __graft_entry__ as g; g.build()') or no llvm-objdump")
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "o.o")
        os.symlink(obj, tmp)
        subprocess.run([LLVM + "llvm-objdump", "--offloading", tmp], capture_output=True, cwd=d, check=True)
        co = [f for f in os.listdir(d) if "amdgcn" in f]
        if not co:
            return "", ""
        path = os.path.join(d, co[0])
        asm = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", path], capture_output=True, text=True, check=True).stdout
    return asm, notes


def _functions(asm):
    """{symbol: [instruction lines]} of a disassembly."""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None and line.strip():
            out[cur].append(line.strip())
    return out


@pytest.mark.parametrize("unit", UNITS)
def test_no_packed_fp32_next_to_mfma(unit):
    asm, _ = _device_code(unit)
    bad = []
    for name, ins in _functions(asm).items():
        if any("v_mfma" in i for i in ins):
            packed = [i for i in ins if re.search(r"\bv_pk_(mul|fma|add)_f32\b", i)]
            if packed:
                bad.append((name, len(packed), packed[0]))
    assert not bad, f"{unit}: kernels with MFMAs AND packed fp32 VALU (build without -fno-slp-vectorize?): {bad[:3]}"


@pytest.mark.parametrize("unit", ["urnn_small", "urnn_coop_tiles", "urnn_elem"])
def test_cooperative_kernels_use_no_scratch(unit):
    _, notes = _device_code(unit)
    rx = re.compile(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", re.S)
    found = [(n, int(p), int(sp)) for n, p, sp in rx.findall(notes) if "coop" in n]
    assert found, f"{unit}: no cooperative kernel found in the object's metadata"
    bad = [f for f in found if f[1] != 0 or f[2] != 0]
    assert not bad, f"cooperative kernels with scratch / spills: {bad}"
