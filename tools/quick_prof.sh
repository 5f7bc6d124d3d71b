#!/bin/bash
# per-kernel average durations of the bench workload (rocprofv3 --kernel-trace --stats), printed as a table
R=$PWD; O=$R/gpurun_out/quick_prof; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile ${BENCH_ARGS} > $O/run.log 2>&1
cd $R
f=$(ls -t $O/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'EOF2'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 13 + 10      # warm-up + timed + native fp32 probe is disabled by --no-profile
print("kernel (calls) avg us | share")
for r in rows[:24]:
    print("%-100s %5d  %8.1f us  %5.1f %%" % (r['Name'][:100], int(r['Calls']), float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
EOF2
grep -o '"value": [0-9.]*' $O/run.log | head -1
