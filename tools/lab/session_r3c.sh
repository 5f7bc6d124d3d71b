#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -4 > $O/tests.txt
for rep in 1 2; do for b in mlp_c96_old mlp_c96_new; do echo -n "$b: "; timeout 120 build/labs/$b; done; done > $O/mlp_labs.txt 2>&1
for b in build/labs/abl96_*; do echo -n "$(basename $b): "; timeout 60 $b; done > $O/abl96.txt 2>&1
cat $O/tests.txt $O/mlp_labs.txt $O/abl96.txt
