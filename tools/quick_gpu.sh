#!/bin/bash
# On the GPU box (via gpurun): the GPU suite, then the bench summary (twice), optionally the A/B of an environment switch.
#   bash tools/quick_gpu.sh [tag] [pytest -k expression | "all" | "none"] [ENV=VALUE for an extra A/B bench]
TAG=${1:-q}; SEL=${2:-all}; AB=$3
if [ "$SEL" = all ]; then python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
elif [ "$SEL" != none ]; then python -m pytest tests -m gpu -x -q -k "$SEL" 2>&1 | grep -E "passed|failed|error" | tail -3; fi
summ() { python - "$1" <<'EOF2'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels"]
print("%.0f clips/s %.3f ms | " % (d["value"], d["ms_per_step"]) + " ".join("%s %.3f" % (n, k[n]["ms_per_step"]) for n in k) +
      " | wide %.3f dw %.3f" % (d["roofline"]["frac"], d["roofline_dwconv"]["frac"]))
EOF2
}
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && summ gpurun_out/${TAG}_bench.json
done
if [ -n "$AB" ]; then
  env $AB python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_ab_bench.json 2> gpurun_out/${TAG}_ab_bench.err && echo "with $AB:" && summ gpurun_out/${TAG}_ab_bench.json
fi
