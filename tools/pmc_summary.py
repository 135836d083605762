"""Reduce the rocprofv3 passes of tools/profile_round.sh to one per-kernel table, keyed by bench.py's kernel names (prof.hip):

    python tools/pmc_summary.py <kernel_stats.csv> <pmc dir> [<pmc dir> ...]  > profiles/<tag>_pmc_summary.json

Per kernel: launches, avg_us (kernel-trace stats pass), hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE is
doubled: gfx950 reports about half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is),
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) - the share of SIMD-cycles in which the
matrix pipe of a SIMD is executing an MFMA, averaged over the chip and the kernel's run time - plus the wave-cycle split
(SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = stalled at issue, SQ_ACTIVE_INST_ANY = issuing) and the MFMA
instruction count. Counters of one kernel are means over its launches; every counter set was collected in its own pass."""
import csv
import glob
import json
import os
import re
import sys


def bench_name(k):
    """rocprof kernel symbol -> bench.py / prof.hip kernel class (None = not one of ours)."""
    if "vitblock_kernel" in k:
        return "vit_block"
    if "vit_stem_kernel" in k:
        return "vit_stem"
    if "vit_qkv_kernel" in k:
        return "vit_qkv"
    if "enc_chain_kernel" in k:
        return "row_chain_enc"
    if "mlp_chain_split_kernel" in k:
        return "row_chain_split"
    if "mlp_chain_kernel" in k:
        return "row_chain_rowwave"
    if "mlp_kernel" in k or "mlp_small_kernel" in k:
        return "mlp_fused"
    if "attn_lds_kernel" in k:
        return "attn_global"
    if "attn_win_kernel" in k:
        m = re.search(r"attn_win_kernelI\w+?Li(\d+)E", k)
        return "attn_window_one_wave_hd%s" % m.group(1) if m else "attn_window_one_wave"
    if "conv3x3_patch_kernel" in k:
        return "conv3x3_patch"
    if "attn_kernel" in k:
        m = re.search(r"attn_kernelI\w+?Li(\d+)ELi(\d+)E", k)
        return "attn_kernel_hd%s_qt%s" % (m.group(1), m.group(2)) if m else "attn_kernel"
    if "gemm_big_kernel" in k:
        return "gemm_mfma_big"
    m = re.search(r"gemm_(?:dma_)?kernelI\w+?Li(\d+)ELi(\d+)ELi(\d)E", k)
    if m:
        return {"0": "gemm_mfma", "1": "gemm_mfma_conv3x3", "2": "gemm_mfma_patch"}[m.group(3)] + "_%sx%s" % (m.group(1), m.group(2))
    for frag, name in (("layernorm_kernel", "layernorm_rows"), ("msda_fused_kernel", "msda_fused_forward"), ("topk_kernel", "topk"),
                       ("postprocess_kernel", "postprocess"), ("decoder_inputs_kernel", "decoder_inputs"),
                       ("select_gather_kernel", "select_gather"), ("rowmax_kernel", "rowmax"), ("box_reparam_kernel", "box_reparam")):
        if frag in k:
            return name
    return None


def main():
    stats, dirs = sys.argv[1], sys.argv[2:]
    table = {}
    for r in csv.DictReader(open(stats)):
        n = bench_name(r["Name"])
        if n is None:
            continue
        e = table.setdefault(n, {"launches": 0, "total_ns": 0.0})
        e["launches"] += int(r["Calls"]); e["total_ns"] += float(r["TotalDurationNs"])
    ctr = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                n = bench_name(r["Kernel_Name"])
                if n is None:
                    continue
                s = ctr.setdefault(n, {}).setdefault(r["Counter_Name"], [0.0, 0])
                s[0] += float(r["Counter_Value"]); s[1] += 1
    out = {"_note": __doc__.split("\n\n")[1].replace("\n", " ")}
    tot = sum(e["total_ns"] for e in table.values())
    for n, e in sorted(table.items(), key=lambda kv: -kv[1]["total_ns"]):
        c = {k: v[0] / v[1] for k, v in ctr.get(n, {}).items()}
        row = {"launches": e["launches"], "avg_us": round(e["total_ns"] / e["launches"] / 1e3, 2),
               "share_of_our_kernel_time": round(e["total_ns"] / tot, 4)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            row["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            row["fetch_kib_raw"], row["write_kib_raw"] = round(c["FETCH_SIZE"], 1), round(c["WRITE_SIZE"], 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0), 4)
        if c.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    row[k.lower() + "_frac_of_wave_cycles"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
        for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum", "TCC_MISS_sum"):
            if k in c:
                row[k] = int(c[k])
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            row["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        out[n] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
