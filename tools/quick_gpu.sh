python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -k "block or e2e or stress" 2>&1 | tail -4
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err
python - <<EOF2
import json
d=json.load(open("gpurun_out/bench_q.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("   %-10s %2d launches %.3f ms" % (k, v["launches_per_step"], v["ms_per_step"]))
EOF2
