#!/bin/bash
# On the GPU box: the bf16 parity suite on the matrix-pipe depthwise kernel, then bf16a bench A/B (ACX_DW_MFMA = 1 | 0).
#   bash tools/lab/dwm_ab.sh [tag]
TAG=${1:-dwm}
export ACX_RECORD_FLOOR=/tmp/floor_$TAG.json
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -6
unset ACX_RECORD_FLOOR
cp /tmp/floor_$TAG.json gpurun_out/floor_$TAG.json 2>/dev/null
for v in 1 0 1 0; do
  ACX_DW_MFMA=$v timeout 300 python bench.py --precision bf16a --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_m$v.json 2> gpurun_out/${TAG}_m$v.err
  python - <<EOF2
import json
d=json.load(open("gpurun_out/${TAG}_m$v.json"))
print("DW_MFMA=$v", round(d["value"]), round(d["ms_per_step"],3), {k:round(x["ms_per_step"],3) for k,x in d["kernels"].items()}, d.get("roofline_dwconv"))
EOF2
done
