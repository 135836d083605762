"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes to HBM bytes per launch of each of our kernels.

FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports about half of a
wide coalesced read stream); WRITE_SIZE is taken as is. Output keys are bench.py's kernel names (prof.hip)."""
import csv
import glob
import json
import os
import sys

NAMES = {"mlp_kernel": "mlp_fused", "attn_kernel": None, "gemm_dma_kernel": "gemm_mfma", "gemm_kernel": "gemm_mfma",
         "layernorm_kernel": "layernorm_rows", "msda_fused_kernel": "msda_fused_forward"}


def per_kernel(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {"_note": "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, rocprofv3 --pmc, separate passes, bench.py "
                    "default workload (LW-DETR-small, B=32, fp16); per-kernel means over all launches of the pass"}
    groups = {}
    for kname, vals in fetch.items():
        w = write.get(kname, [0.0])
        key = None
        for frag, label in NAMES.items():
            if frag in kname:
                key = label
                if frag == "attn_kernel":
                    key = "attn_global" if "Li16ELi4" in kname else ("attn_window" if "Li16ELi2" in kname else "attn_decoder")
        if key is None:
            continue
        g = groups.setdefault(key, {"fetch": [], "write": []})
        g["fetch"] += vals
        g["write"] += w
    for key, g in groups.items():
        f, w = sum(g["fetch"]) / len(g["fetch"]), sum(g["write"]) / max(1, len(g["write"]))
        out[key] = {"bytes_per_launch": int((2 * f + w) * 1024), "fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1),
                    "launches": len(g["fetch"])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
