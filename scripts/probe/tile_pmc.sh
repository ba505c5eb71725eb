cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf /tmp/px; mkdir -p /tmp/px
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/px -- python3 scripts/c5_batched_ip.py 2000000 > /tmp/px/out.txt 2>/tmp/px/err.txt
f=$(find /tmp/px -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    for k in ("block_dist_bf16x3_kernel","hnsw_search_kernel","block_merge_kernel"):
        if k in n:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in acc.items():
    d=sorted(dur[k])[len(dur[k])//2]
    print(k, "median launch ns", d, {c: round(sorted(x)[len(x)//2]) for c,x in v.items()})
PY
tail -3 /tmp/px/err.txt
