#!/usr/bin/env python
"""profiles/pmc_kernels.json (inference step) / profiles/pmc_train.json (training step) from the FETCH_SIZE / WRITE_SIZE summaries of
tools/collect_profiles.sh (tools/pmc_summary.py text): counter traffic per launch of every kernel family bench.py reports a roofline
for, stamped with the hash of the kernel sources (and of the .so, and the box) that produced them -- bench.py prints `traffic`
only when the source hash matches its own tree.
usage (on the GPU box, from collect_profiles.sh): python tools/make_pmc_json.py <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> <out.json> [tag]"""
import hashlib
import json
import os
import re
import socket
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (kernel_source_hash: the same function bench.py checks the record with)
FAMILIES = {   # bench.py's roofline families -> kernel-name pattern (template arguments: NB, PB, MAP, EPI, ...)
    "gates": re.compile(r"conv_gemm_kernel<\s*\d+,\s*\d+,\s*\d+,\s*3,"),                 # EPI_GRU1 = 3, the >= 24 000-pixel planes
    "candidate": re.compile(r"cand_fused_kernel<|cand_gated_kernel<|conv_gemm_kernel<\s*\d+,\s*\d+,\s*\d+,\s*4,"),   # fused / EPI_CAND = 4
    "blend": re.compile(r"gru_blend_kernel<\s*4,"),                                     # the 16-byte form: full and half resolution at 500x500
    "head": re.compile(r"head_k[1-4]<"),
    "small_cells": re.compile(r"small_cell_gemm_kernel<|coop_cell_kernel<"),
    "wgrad": re.compile(r"wgrad_kernel<"),
}


def lib_hash(path=None):
    path = path or os.environ.get("URNN_LIB") or os.path.join(REPO, "u-rnn_amd", "liburnn_hip.so")
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def parse(path, counter):
    """-> ({kernel: (dispatches, value per dispatch)}, total per frame or None)"""
    per, total, name, n = {}, None, None, 0
    for line in open(path):
        m = re.match(r"(\S.*?)\s+dispatches=(\d+)\s+avg_us", line)
        if m:
            name, n = m.group(1), int(m.group(2))
            continue
        m = re.match(r"\s+%s\s+([\d.]+)" % counter, line)
        if m and name:
            per[name] = (n, float(m.group(1)))
            continue
        m = re.match(r"#\s+%s\s+([\d.]+)" % counter, line)
        if m:
            total = float(m.group(1))
    return per, total


def main(fetch_txt, write_txt, out, tag=""):
    fe, fe_tot = parse(fetch_txt, "FETCH_SIZE")
    wr, wr_tot = parse(write_txt, "WRITE_SIZE")
    fams = {}
    for fam, pat in FAMILIES.items():
        ks = [k for k in fe if pat.search(k) and k in wr]
        if not ks:
            continue
        nd = sum(fe[k][0] for k in ks)
        fetch = sum(fe[k][0] * fe[k][1] for k in ks) / max(nd, 1)
        write = sum(wr[k][0] * wr[k][1] for k in ks) / max(sum(wr[k][0] for k in ks), 1)
        fams[fam] = {"dispatches": nd, "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                     "instantiations": {k: {"dispatches": fe[k][0], "FETCH_SIZE_KiB_raw": fe[k][1], "WRITE_SIZE_KiB_raw": wr[k][1]} for k in ks}}
    rec = {
        "families": fams,
        "note": "per family: dispatch-weighted average over its launches in the profiled frames (at 500x500: the full- AND half-resolution "
                "launches together, like bench.py's bytes_per_launch)",
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                      "WRITE_SIZE as reported; counters are KiB (x1024); separate --pmc passes",
        "source": f"{os.path.basename(fetch_txt)} + {os.path.basename(write_txt)} ({tag})",
        "kernel_source_sha256": bench.kernel_source_hash(),
        "lib_sha256": lib_hash(),
        "box": socket.gethostname(),
    }
    if fe_tot and wr_tot:
        rec["whole_step"] = {"FETCH_SIZE_KiB_per_frame": fe_tot, "WRITE_SIZE_KiB_per_frame": wr_tot,
                             "hbm_bytes_per_frame": (2.0 * fe_tot + wr_tot) * 1024.0,
                             "note": "one-chain eager schedule so that every dispatch is attributed; FETCH doubled as above"}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
