#!/bin/bash
# On the GPU box: the bf16 parity suite on the paired fused MLP, then bf16a bench A/B (ACX_BF16_PAIR = 1 | 0) and per-kernel durations.
#   bash tools/lab/pair_ab.sh [tag]
TAG=${1:-pair}
R=$PWD
export ACX_RECORD_FLOOR=/tmp/floor_$TAG.json
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -4
unset ACX_RECORD_FLOOR
for v in 1 0; do
  ACX_BF16_PAIR=$v timeout 300 python bench.py --precision bf16a --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_p$v.json 2> gpurun_out/${TAG}_p$v.err
  python - <<EOF2
import json
d=json.load(open("gpurun_out/${TAG}_p$v.json"))
print("PAIR=$v", round(d["value"]), round(d["ms_per_step"],3), {k:round(x["ms_per_step"],3) for k,x in d["kernels"].items()}, round(d["roofline"]["frac"],3))
EOF2
done
export TMPDIR=/tmp; cd /tmp
ACX_BF16_PAIR=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -- python3 $R/tools/prof_step.py --precision bf16a --steps 3 > /dev/null 2>&1
f=$(ls -t $(find $R/gpurun_out/${TAG}_kt -name "*kernel_stats.csv") | head -1); grep -E "mlp_" $f | cut -c1-70,160-230
