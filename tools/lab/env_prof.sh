#!/bin/bash
# On the GPU box: per-kernel average durations (rocprofv3 --kernel-trace --stats) of the bf16a bench, one stream, for each value of
# an environment switch.   bash tools/lab/ns_prof.sh VAR v1 v2 ...
VAR=${1:?environment switch, e.g. ACX_DW_MFMA}; shift; VALS=${@:-0 1}
R=$PWD; export TMPDIR=/tmp
for v in $VALS; do
  O=$R/gpurun_out/ns_prof_$v; rm -rf $O; mkdir -p $O
  export $VAR=$v ACX_SPLIT_STREAMS=0
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --precision bf16a --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-extra-configs > $O/run.log 2>&1
  cd $R
  f=$(ls -t $O/*/*kernel_stats.csv | head -1)
  echo "== $VAR=$v  $(grep -o '"value": [0-9.]*' $O/run.log | head -1)"
  python3 - "$f" <<'EOF2'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s %5d  %8.1f us" % (r['Name'][:90], int(r['Calls']), float(r['AverageNs']) / 1e3))
EOF2
done
