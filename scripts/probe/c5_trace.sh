#!/bin/bash
# per-launch durations of one configs[4] batch (the last of the run): rocprofv3 kernel trace of scripts/c5_batched_ip.py
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=/tmp/c5t_$$_$RANDOM; mkdir -p $T; cd $R
rocprofv3 --kernel-trace --output-format csv -d $T -- python3 scripts/c5_batched_ip.py 10000000 > /dev/null 2> $T/err.log
f=$(find $T -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
sel=[r for r in rows if any(k in r['Kernel_Name'] for k in ('p1_tile','block_merge','block_final','block_rescore','p8_round','p1_eps','select'))]
sel.sort(key=lambda r:int(r['Start_Timestamp']))
# the last batch: from the last p8_round_queries on
idx=[i for i,r in enumerate(sel) if 'p8_round' in r['Kernel_Name']]
start=idx[-2] if len(idx)>1 else 0
t0=int(sel[start]['Start_Timestamp'])
prev_end=t0
for r in sel[start:idx[-1] if len(idx)>1 else None]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  +gap {(s-prev_end)/1e3:6.1f}  dur {(e-s)/1e3:8.1f} us  grid {r['Grid_Size_X']:>8}  {r['Kernel_Name'][:60]}")
    prev_end=e
PY
