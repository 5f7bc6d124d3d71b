#!/bin/bash
# On the GPU box: fp32_split bench A/B of the paired fused MLP of stages 0-1 (ACX_SPLIT_PAIR = 1 | 0), twice, then per-kernel durations.
TAG=${1:-sp}
R=$PWD
for rep in 1 2; do for v in 2 1 0; do
  ACX_SPLIT_PAIR=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${TAG}_p$v.json 2> gpurun_out/${TAG}_p$v.err
  python - <<EOF2
import json
d=json.load(open("gpurun_out/${TAG}_p$v.json"))
print("SPLIT_PAIR=$v", round(d["value"]), round(d["ms_per_step"],3), {k:round(x["ms_per_step"],3) for k,x in d["kernels"].items()}, "wide", round(d["roofline"]["frac"],3), "all", round(d["roofline_all_pointwise"]["frac"],3))
EOF2
done; done
export TMPDIR=/tmp; cd /tmp
ACX_SPLIT_PAIR=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -- python3 $R/tools/prof_step.py --precision fp32_split --steps 3 > /dev/null 2>&1
f=$(ls -t $(find $R/gpurun_out/${TAG}_kt -name "*kernel_stats.csv") | head -1); grep -E "mlp_" $f | cut -c1-80,150-230
