#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_pass.sh into the files kept under profiles/:
  <dir>/kernel_stats.csv   the --stats table (haet kernels + the rest)
  <dir>/pmc.json           per kernel: median FETCH_SIZE / WRITE_SIZE per launch (KiB as reported; FETCH_SIZE must be
                           doubled on gfx950 for wide coalesced reads, MI355X_MICROARCH.md HBM section) and the SQ
                           counters per launch with derived MFMA utilisation.
Usage: profile_summary.py gpurun_out/prof_<tag> [points]"""
import csv
import glob
import json
import os
import statistics
import sys

SHORT = (("conv8_kernel", "conv8"), ("presplit_kernel", "conv_presplit"), ("gemm_ws_kernel<8, 4, true", "gemm_ws_split"), ("gemm_ws_kernel", "gemm_ws"), ("deslice_out_kernel", "deslice_out"),
         ("deslice_mfma_kernel", "deslice"), ("slice_mfma_kernel", "slice"), ("slice_merge_kernel", "merge"),
         ("erwin2_kernel", "token_stage"), ("erwin_kernel<true>", "token_stage_v1"), ("erwin_kernel", "erwin"),
         ("slice_bwd_x3_kernel", "slice_bwd_x3"), ("balltree_wave_kernel", "balltree_wave"),
         ("balltree_kernel", "balltree"))


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return None


def main():
    d = sys.argv[1]
    res = {"kernels": {}}
    stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(d, "kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:24]:
                w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        for r in rows:
            s = short(r["Name"])
            if s and s not in res["kernels"]:
                res["kernels"][s] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])}
    for pdir in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(pdir):
            continue
        for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                s = short(r.get("Kernel_Name", ""))
                if not s:
                    continue
                key = (s, r["Counter_Name"], r.get("Dispatch_Id", r.get("Correlation_Id", "")))
                per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
            byk = {}
            for (s, c, _), v in per.items():
                byk.setdefault((s, c), []).append(v)
            for (s, c), vals in byk.items():
                res["kernels"].setdefault(s, {})[c] = statistics.median(vals)
    for s, k in res["kernels"].items():
        if "FETCH_SIZE" in k:
            k["fetch_kb_raw"] = k.pop("FETCH_SIZE")
        if "WRITE_SIZE" in k:
            k["write_kb"] = k.pop("WRITE_SIZE")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "SQ_BUSY_CYCLES" in k and k["SQ_BUSY_CYCLES"]:
            # MFMA-busy cycles summed over the SIMDs that ran waves / (busy cycles x 4 SIMDs per CU), per the counters'
            # own units (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES per SE/XCD aggregate)
            k["mfma_busy_over_sq_busy"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CYCLES"]
        if "SQ_INSTS_MFMA" in k and "avg_us" in k:
            k["mfma_insts_per_us"] = k["SQ_INSTS_MFMA"] / k["avg_us"]
    json.dump(res, open(os.path.join(d, "pmc.json"), "w"), indent=1)
    # bench.py's roofline.traffic source: FETCH_SIZE / WRITE_SIZE per kernel of THIS build (source hash)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    points = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    alg = {"gemm_ws_split": 2 * 1024, "gemm_ws": 2 * 1024, "slice": 1024, "deslice": 2 * 1024, "deslice_out": 2 * 1024}
    traffic = {"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py "
                           "--steps 3 --warmup 1 --no-cpu-baseline; median per launch, KiB as reported.  On gfx950 FETCH_SIZE "
                           "reports 1/2 of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM): corrected = 2 x raw; "
                           "WRITE_SIZE is exact.", "points": points, "source_id": bench.kernel_source_id(), "kernels": {}}
    for s_, k in res["kernels"].items():
        if "fetch_kb_raw" in k and "write_kb" in k:
            traffic["kernels"][s_] = {"fetch_kb_raw": k["fetch_kb_raw"], "write_kb": k["write_kb"]}
            if s_ in alg:
                traffic["kernels"][s_]["algorithmic_kb"] = alg[s_] * points / 1024.0
    json.dump(traffic, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
    for s, k in res["kernels"].items():
        print(s, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in k.items()}))


if __name__ == "__main__":
    main()
