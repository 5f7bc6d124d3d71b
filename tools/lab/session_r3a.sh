#!/bin/bash
# round 3, GPU session a: co-issue table, old-vs-new fused MLP labs, parity + stress suites, quick bench
O=gpurun_out/r3a; mkdir -p $O
timeout 300 build/labs/coissue_bench > $O/coissue.txt 2>&1
for rep in 1 2; do
  for b in mlp_c96_old mlp_c96_new mlp_c192_old mlp_c192_new mlp_c384_old mlp_c384_new; do echo -n "$b: "; timeout 120 build/labs/$b; done
done > $O/mlp_labs.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -8 > $O/tests.txt
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/coissue.txt $O/mlp_labs.txt $O/tests.txt; cut -c1-600 $O/bench.json
