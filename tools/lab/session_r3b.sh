#!/bin/bash
# round 3, GPU session b: persistent stage-0 kernel -- parity first, then lab + bench
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -8 > $O/tests.txt
for rep in 1 2; do
  for b in mlp_c96_old mlp_c96_new; do echo -n "$b: "; timeout 120 build/labs/$b; done
done > $O/mlp_labs.txt 2>&1
for m in 65536 131072 262144; do echo -n "mlp_c96_new M=$m: "; timeout 60 build/labs/mlp_c96_new $m; done >> $O/mlp_labs.txt 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/tests.txt $O/mlp_labs.txt; python - <<EOF2
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("   %-10s %2d launches %.3f ms" % (k, v["launches_per_step"], v["ms_per_step"]))
EOF2
