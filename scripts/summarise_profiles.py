#!/usr/bin/env python3
"""Reduce rocprofv3 output directories to the small files kept under profiles/ (runs on the GPU box,
because the raw traces of a 10M-vector build exceed what gpurun copies back).

    python3 scripts/summarise_profiles.py --stats-dir D1 --fetch-dir D2 --write-dir D3 \
        --bench-json B.json --out OUTDIR --tag r01_h

Writes  <tag>_kernel_stats.csv           (rocprofv3 --stats, verbatim)
        <tag>_pmc_FETCH_SIZE_hnsw_search.csv / <tag>_pmc_WRITE_SIZE_hnsw_search.csv
                                          (counter rows of the search kernel's TIMED launches only)
        <tag>_traffic.json               (HBM bytes per launch, corrected as MI355X_MICROARCH.md 'HBM' prescribes:
                                          FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced reads, KB -> x1024, WRITE_SIZE exact)
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["hnsw_device.hpp", "kernels_arith.hip", "kernels_misc.hip", "kernels.hpp"]  # what the fused search / insert / exact kernels are built from


def kernel_sources_sha16():
    """What a traffic record is valid for: the kernel sources it was measured on (bench.py recomputes this and reports a
    record as stale -- roofline.traffic null -- when they have changed since)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "vector_store_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def search_rows(path, counter, kernel="hnsw_search_kernel"):
    """Counter rows of one kernel's launches with its largest grid (search: the bench batch; insert: full sub-batches)."""
    rows = []
    with open(path, newline="") as f:
        r = csv.DictReader(f)
        for row in r:
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                rows.append(row)
        fields = r.fieldnames
    if not rows:
        return fields, []
    g = max(int(x["Grid_Size"]) for x in rows)
    return fields, [x for x in rows if int(x["Grid_Size"]) == g]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats-dir")
    ap.add_argument("--fetch-dir")
    ap.add_argument("--write-dir")
    ap.add_argument("--tcc-dir", help="a pass with --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum (one pass: four TCC slots)")
    ap.add_argument("--bench-json", help="JSON line of the bench run made under --stats-dir (workload string, algorithmic bytes)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--timed-launches", type=int, default=3, help="PMC passes: the last N launches are the timed steps")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.stats_dir:
        ks = find(a.stats_dir, "_kernel_stats.csv")
        if ks:
            shutil.copy(ks, os.path.join(a.out, f"{a.tag}_kernel_stats.csv"))
    raw = {}
    kernel = None
    for name, d in (("FETCH_SIZE", a.fetch_dir), ("WRITE_SIZE", a.write_dir)):
        if not d:
            continue
        cc = find(d, "_counter_collection.csv")
        if not cc:
            continue
        fields, rows = search_rows(cc, name)
        rows = rows[-a.timed_launches:]
        with open(os.path.join(a.out, f"{a.tag}_pmc_{name}_hnsw_search.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=fields)
            w.writeheader()
            w.writerows(rows)
        if rows:
            raw[name] = sum(float(x["Counter_Value"]) for x in rows) / len(rows)
            kernel = rows[0]["Kernel_Name"]
    if "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
        rec = {"kernel": kernel,
               "raw": {"FETCH_SIZE_KB_per_launch": raw["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": raw["WRITE_SIZE"]},
               "correction": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16 B/lane "
                             "coalesced reads -> x2; WRITE_SIZE exact; counters are in KB -> x1024; separate --pmc passes",
               "hbm_bytes_per_launch": 2 * raw["FETCH_SIZE"] * 1024 + raw["WRITE_SIZE"] * 1024}
        if a.bench_json and os.path.exists(a.bench_json):
            for line in open(a.bench_json):
                line = line.strip()
                if line.startswith("{"):
                    b = json.loads(line)
                    nq = int(b["config"]["workload"].split(",")[1].split()[0])
                    rec["workload"] = b["config"]["workload"]
                    rec["algorithmic_bytes_per_launch"] = b["roofline"]["bytes_per_query"] * nq
                    rec["ratio_traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
        # How much of the traffic is served by the XCDs' L2s and how much leaves them (MI355X_MICROARCH.md 'L2': hit rate =
        # TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); TCC_EA0_RDREQ = read requests on the L2's memory side, _DRAM = those that
        # target HBM (behind the Infinity Cache, whose hits no counter of this list separates)
        if a.tcc_dir:
            cc = find(a.tcc_dir, "_counter_collection.csv")
            tcc = {}
            for name in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum"):
                if not cc:
                    break
                _, rows = search_rows(cc, name)
                rows = rows[-a.timed_launches:]
                if rows:
                    tcc[name] = sum(float(x["Counter_Value"]) for x in rows) / len(rows)
            if len(tcc) == 4:
                hit, miss = tcc["TCC_HIT_sum"], tcc["TCC_MISS_sum"]
                rec["dram"] = {"counters_per_launch": tcc, "l2_hit_rate": hit / max(hit + miss, 1.0),
                               "l2_requests_served_in_l2_frac": hit / max(hit + miss, 1.0),
                               "memory_side_read_requests_to_dram_frac": tcc["TCC_EA0_RDREQ_DRAM_sum"] / max(tcc["TCC_EA0_RDREQ_sum"], 1.0),
                               "bytes_beyond_l2_per_launch_64B_per_request_x2": tcc["TCC_EA0_RDREQ_sum"] * 64 * 2,
                               "note": "L2 hits are served on the XCD; every memory-side read request (TCC_EA0_RDREQ, tallied at 64 B for 128-B requests on "
                                       "gfx950: x2) leaves the XCD for the Infinity Cache / HBM; no counter of rocprofv3 -L separates Infinity-Cache hits from "
                                       "HBM reads, so 'beyond L2' is the upper bound of the DRAM traffic"}
        rec["kernel_sources_sha16"] = kernel_sources_sha16()
        json.dump(rec, open(os.path.join(a.out, f"{a.tag}_traffic.json"), "w"), indent=1)
    # the other two hot kernels of the same run: the build's insert kernel (HBM-bound) and the exact path's MFMA tile kernel
    bench = None
    if a.bench_json and os.path.exists(a.bench_json):
        for line in open(a.bench_json):
            if line.strip().startswith("{"):
                bench = json.loads(line)
    stats = {}
    ks = find(a.stats_dir, "_kernel_stats.csv") if a.stats_dir else None
    if ks:
        for row in csv.DictReader(open(ks, newline="")):
            stats[row["Name"]] = row
    for kernel, label in (("hnsw_insert_kernel", "insert"), ("exact_dist_mfma_kernel", "mfma")):
        raw2, kname = {}, None
        for name, d in (("FETCH_SIZE", a.fetch_dir), ("WRITE_SIZE", a.write_dir)):
            cc = find(d, "_counter_collection.csv") if d else None
            if not cc:
                continue
            _, rows = search_rows(cc, name, kernel)
            if rows:
                raw2[name] = sum(float(x["Counter_Value"]) for x in rows) / len(rows)
                raw2[name + "_launches"] = len(rows)
                raw2[name + "_ms"] = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows) / len(rows) / 1e6
                raw2["grid"] = int(rows[0]["Grid_Size"])
                kname = rows[0]["Kernel_Name"]
        if "FETCH_SIZE" not in raw2 or "WRITE_SIZE" not in raw2:
            continue
        rec = {"kernel": kname, "launches_averaged": raw2["FETCH_SIZE_launches"], "grid_threads": raw2["grid"],
               "raw": {"FETCH_SIZE_KB_per_launch": raw2["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": raw2["WRITE_SIZE"]},
               "hbm_bytes_per_launch": 2 * raw2["FETCH_SIZE"] * 1024 + raw2["WRITE_SIZE"] * 1024,
               "kernel_sources_sha16": kernel_sources_sha16()}
        rec["full_grid_launch_ms_in_the_pmc_pass"] = raw2["FETCH_SIZE_ms"]
        st = stats.get(kname)
        if st:
            rec["rocprof_average_ms_all_launches"] = float(st["AverageNs"]) / 1e6
            rec["rocprof_max_ms"] = float(st["MaxNs"]) / 1e6
        if bench and label == "insert" and "build" in bench:
            b = bench["build"]
            nodes = raw2["grid"] // 64
            row_b = bench["roofline"]["bytes_per_query"] and (bench["roofline"]["bytes_per_query"] - bench["roofline"]["hops_per_query"] * 132
                                                              - int(bench["config"]["workload"].split("x")[1].split()[0]) * 4) / bench["roofline"]["evals_per_query"]
            if "insert_evals_per_add" in b:
                rec["algorithmic_bytes_per_launch"] = nodes * (b["insert_evals_per_add"] * row_b + b["hops_per_add"] * 132)
                rec["ratio_traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
                rec["achieved_TBps"] = rec["algorithmic_bytes_per_launch"] / (raw2["FETCH_SIZE_ms"] * 1e-3) / 1e12
                rec["frac_of_8TBps"] = rec["achieved_TBps"] / 8.0
        if bench and label == "mfma":
            dim = int(bench["config"]["workload"].split("x")[1].split()[0])
            tiles = raw2["grid"] // 256  # 128 x 128 score tiles, one 256-thread workgroup each
            rec["flops_per_launch"] = 2.0 * tiles * 128 * 128 * dim
            rec["achieved_TFLOPs"] = rec["flops_per_launch"] / (raw2["FETCH_SIZE_ms"] * 1e-3) / 1e12
            rec["frac_of_157_TFLOPs_f32_mfma"] = rec["achieved_TFLOPs"] / 157.3
            rec["algorithmic_bytes_per_launch"] = (tiles ** 0.5 * 128 * 2) * dim * 4  # lower bound: a square block of rows + queries once
        json.dump(rec, open(os.path.join(a.out, f"{a.tag}_traffic_{label}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
