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
import json
import os
import shutil


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def search_rows(path, counter):
    """Counter rows of the hnsw_search kernel with the bench batch's grid (the largest grid seen)."""
    rows = []
    with open(path, newline="") as f:
        r = csv.DictReader(f)
        for row in r:
            if "hnsw_search_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
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
        json.dump(rec, open(os.path.join(a.out, f"{a.tag}_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
