#!/usr/bin/env python3
"""Reduce scripts/profile_c5.sh's rocprofv3 directories to <tag>_c5_kernel_stats.csv and <tag>_c5_kernels.json: per kernel
of the exact q = 256 path -- launches, average duration, HBM bytes per launch (FETCH_SIZE x 2 on gfx950, WRITE_SIZE, KB ->
x 1024, separate --pmc passes: MI355X_MICROARCH.md 'HBM'), and for the split-bf16 tile kernel the matrix-pipe occupancy
(SQ_VALU_MFMA_BUSY_CYCLES over duration x SIMDs) and its rate against the dense bf16 peak."""
import argparse
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarise_profiles import kernel_sources_sha16  # noqa: E402

KERNELS = ("p1_tile_kernel", "p1_plane_rows_kernel", "p1_round_queries_kernel", "p8_plane_rows_kernel", "p8_round_queries_kernel", "block_merge512_kernel", "block_dist_bf16x3_kernel", "block_merge_kernel", "exact_select_kernel", "exact_finish_kernel", "block_rescore_kernel", "block_final_kernel",
           "split_queries_kernel", "exact_dist_mfma_kernel")


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def per_kernel(path, counter):
    """kernel -> (launches, mean counter value, mean duration ms, grid, workgroup) over the launches with the largest grid."""
    rows = {}
    for row in csv.DictReader(open(path, newline="")):
        if row["Counter_Name"] != counter:
            continue
        for k in KERNELS:
            if k in row["Kernel_Name"]:
                rows.setdefault(k, []).append(row)
    out = {}
    for k, rs in rows.items():
        g = max(int(r["Grid_Size"]) for r in rs)
        rs = [r for r in rs if int(r["Grid_Size"]) == g]
        if k == "p1_tile_kernel":  # persistent launches share one grid: keep the long ones (the last chunk of every batch)
            longest = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
            rs = [r for r in rs if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 0.6 * longest]
        out[k] = (len(rs), sum(float(r["Counter_Value"]) for r in rs) / len(rs),
                  sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / len(rs) / 1e6, g, int(rs[0]["Workgroup_Size"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--vectors", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--plane", default="auto", choices=["auto", "int8", "bf16"], help="which plane the tile kernel streamed (auto: int8 when its instance shows in the stats)")
    a = ap.parse_args()
    stats = {}
    ks = find(os.path.join(a.dir, "stats"), "_kernel_stats.csv")
    if a.plane == "auto":
        a.plane = "int8" if ks and "p1_tile_kernel<false, true>" in open(ks).read() else "bf16"
    if ks:
        shutil.copy(ks, os.path.join(a.out, f"{a.tag}_c5_kernel_stats.csv"))
        for row in csv.DictReader(open(ks, newline="")):
            for k in KERNELS:
                if k in row["Name"]:  # (template instances of one kernel -- the first block's all-scores form, the thresholded form -- add up)
                    e = stats.setdefault(k, {"calls": 0, "total_ms": 0.0})
                    e["calls"] += int(row["Calls"])
                    e["total_ms"] += float(row["TotalDurationNs"]) / 1e6
                    e["average_ms"] = e["total_ms"] / e["calls"]
    pmc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
        cc = find(os.path.join(a.dir, "pmc_" + c), "_counter_collection.csv")
        pmc[c] = per_kernel(cc, c) if cc else {}
    rec = {"workload": f"{a.vectors}x{a.dim} ip, batches of 256 queries (scripts/c5_batched_ip.py)", "kernel_sources_sha16": kernel_sources_sha16(),
           "correction": "FETCH_SIZE x 2 (gfx950, 16 B/lane coalesced reads), WRITE_SIZE exact, KB -> x 1024; one counter per rocprofv3 pass",
           "kernels": {}}
    for k in KERNELS:
        if k not in stats and k not in pmc["FETCH_SIZE"]:
            continue
        e = dict(stats.get(k, {}))
        if k in pmc["FETCH_SIZE"] and k in pmc["WRITE_SIZE"]:
            n, f, ms, grid, wg = pmc["FETCH_SIZE"][k]
            e.update({"full_grid_launches_in_the_pmc_pass": n, "grid_threads": grid, "workgroup": wg, "launch_ms_in_the_pmc_pass": ms,
                      "hbm_bytes_per_launch": 2 * f * 1024 + pmc["WRITE_SIZE"][k][1] * 1024})
        if k == "block_dist_bf16x3_kernel" and "grid_threads" in e:
            tiles = e["grid_threads"] // e["workgroup"]          # one 128 (queries) x 128 (rows) score tile per workgroup
            mfma = 3 * tiles * 128 * 128 * a.dim // (32 * 32 * 16)  # v_mfma_f32_32x32x16_bf16: hi*hi, hi*lo, lo*hi
            e["tiles"] = tiles
            e["mfma_instructions_per_launch"] = mfma
            e["bf16_flops_per_launch"] = mfma * 2 * 32 * 32 * 16
            e["algorithmic_bytes_per_launch"] = (tiles // 2) * 128 * a.dim * 4 + 256 * a.dim * 4  # every row of the block once (two query tiles share it) + the queries
            ms = e.get("average_ms", e["launch_ms_in_the_pmc_pass"])
            e["achieved_bf16_TFLOPs"] = e["bf16_flops_per_launch"] / (ms * 1e-3) / 1e12
            e["frac_of_2500_TFLOPs_dense_bf16"] = e["achieved_bf16_TFLOPs"] / 2500.0
            e["achieved_hbm_TBps_algorithmic"] = e["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e12
            if "hbm_bytes_per_launch" in e:
                e["ratio_traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
            if k in pmc["SQ_VALU_MFMA_BUSY_CYCLES"]:
                _, busy, bms, _, _ = pmc["SQ_VALU_MFMA_BUSY_CYCLES"][k]
                e["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] = busy
                e["expected_busy_cycles_32_per_mfma"] = mfma * 32
                e["launch_ms_in_the_mfma_pass"] = bms
        if k == "p1_tile_kernel" and "grid_threads" in e:
            # the persistent launch with the largest grid that runs longest is the last chunk: rows [32^2 * 1024, N) of the plane
            rows = a.vectors - 1048576 if a.vectors > 1048576 else a.vectors
            int8 = a.plane == "int8"
            kp = (a.dim + 127) // 128 * 128 if int8 else (a.dim + 63) // 64 * 64
            e["plane"] = a.plane
            e["rows_of_the_largest_launch"] = rows
            # ONE product per score: v_mfma_i32_16x16x64_i8 over the 8-bit plane (round 6) / v_mfma_f32_16x16x32_bf16 over the bf16 plane
            e["mfma_instructions_per_launch"] = 256 * rows * kp // (16 * 16 * (64 if int8 else 32))
            e["bf16_flops_per_launch"] = 2 * 256 * rows * kp  # (int8 plane: integer multiply-adds, counted the same way)
            e["algorithmic_bytes_per_launch"] = (rows * (kp + 4) + 256 * kp) if int8 else (rows * kp * 2 + 256 * kp * 2)  # the plane's rows (+ scales) once + the query block once
            ms = max(e.get("launch_ms_in_the_pmc_pass", 0.0), 1e-9)
            if k in pmc["FETCH_SIZE"]:
                ms = pmc["FETCH_SIZE"][k][2]
            e["achieved_bf16_TFLOPs"] = e["bf16_flops_per_launch"] / (ms * 1e-3) / 1e12
            e["frac_of_2500_TFLOPs_dense_bf16"] = e["achieved_bf16_TFLOPs"] / 2500.0
            if a.plane == "int8":
                e["frac_of_5000_TOPs_dense_int8"] = e["achieved_bf16_TFLOPs"] / 5000.0
            e["achieved_hbm_TBps_algorithmic"] = e["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e12
            e["frac_of_8_TBps"] = e["achieved_hbm_TBps_algorithmic"] / 8.0
            if "hbm_bytes_per_launch" in e:
                e["ratio_traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
            if k in pmc["SQ_VALU_MFMA_BUSY_CYCLES"]:
                _, busy, bms, _, _ = pmc["SQ_VALU_MFMA_BUSY_CYCLES"][k]
                e["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] = busy
                e["expected_busy_cycles_16_per_mfma"] = e["mfma_instructions_per_launch"] * 16  # (16 passes of 4 cycles... per 16x16 product, either type)
                e["launch_ms_in_the_mfma_pass"] = bms
        rec["kernels"][k] = e
    json.dump(rec, open(os.path.join(a.out, f"{a.tag}_c5_kernels.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
