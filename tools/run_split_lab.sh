#!/bin/bash
# builds and runs the split-GEMM lab variants on the GPU box: tools/run_split_lab.sh "" -DACX_SLAB_NO_EPI ...
mkdir -p gpurun_out
for v in "$@"; do
  echo "=== variant: [$v]"
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $v tools/split_lab.hip -o /tmp/split_lab && /tmp/split_lab
done 2>&1 | tee gpurun_out/split_lab.txt
