#!/bin/bash
# tools/run_lab.sh <lab source> <variant flags>...   -- builds and runs lab variants on the GPU box
src=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  echo "=== $src variant: [$v]"
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $v $src -o /tmp/lab_bin && /tmp/lab_bin
done 2>&1 | tee -a gpurun_out/lab.txt
