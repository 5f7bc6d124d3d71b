for v in 8 4 6 9 8 5; do
  ACX_DWM_WAVES=$v timeout 300 python bench.py --precision bf16a --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > /tmp/w.json 2>/dev/null
  python - <<EOF2
import json
d=json.load(open("/tmp/w.json"))
print("DWM_WAVES=$v", round(d["value"]), round(d["ms_per_step"],3), "dwconv one-stream", round(d["kernels"]["dwconv"]["ms_per_step"],3))
EOF2
done
