#!/bin/bash
# round 5, GPU session 13: where the single-image graph replay's 0.867 ms go - kernel durations (rocprofv3 kernel trace of the replays) against the wall time
set -u
O=gpurun_out/r5s13; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/tools/lat_bs1.py --n 200 > $R/$O/lat_under_rocprof.txt 2> $R/$O/err.txt
cd $R
cat $O/lat_under_rocprof.txt | grep -v amdgpu
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_bs1.csv
python - <<'PY'
import csv,sys,glob
f=glob.glob('gpurun_out/r5s13/kernel_stats_bs1.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=0; n=0
out=[]
for r in rows:
    calls=int(r['Calls']); avg=float(r['AverageNs']); 
    # replays: 220 (200 + 20 warm) + capture/warm-up calls: per-replay count = round(calls/220)
    per=round(calls/223.0)
    out.append((avg*per/1e3, per, avg/1e3, r['Name'][:90]))
    tot+=avg*per/1e3; n+=per
out.sort(reverse=True)
for o in out[:28]: print(f"{o[0]:8.1f} us/replay  x{o[1]:2d}  avg {o[2]:7.2f} us  {o[3]}")
print("sum of kernel durations per replay: %.1f us over %d launches" % (tot, n))
PY
rm -rf $O/prof
