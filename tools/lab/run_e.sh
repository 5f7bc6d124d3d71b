#!/bin/bash
# bf16 suite on the new ring kernels, then same-box A/B against the library of the commit before (build/labs/libacx_r06base.so)
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -x -q -p no:cacheprovider 2>&1 | tail -3
bash tools/lab/ab_lib.sh build/labs/libacx_r06base.so bf16a bf16
