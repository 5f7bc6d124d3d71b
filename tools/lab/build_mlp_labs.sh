#!/bin/bash
# Builds the stand-alone timing labs of the fused MLP kernels HERE (hipcc cross-compiles; the binaries travel to the GPU box
# under build/labs/): the product sources, and -- when build/labs/old/ holds a copy of an earlier version -- that one too.
#   tools/lab/build_mlp_labs.sh && gpurun -- 'for b in build/labs/mlp_*; do echo $b; $b; done'
set -e
cd "$(dirname "$0")/../.."
S=$PWD/audioset-convnext-inf_amd/csrc; O=build/labs; mkdir -p $O
build() {   # name, source, C, launcher, extra flags
  hipcc -O3 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 --offload-arch=gfx950 -w -DWIDE_C=$3 -DWIDE_FN=$4 -DWIDE_SRC="\"$2\"" $5 tools/wide_lab.hip -o $O/$1 &
}
build mlp_c96_new  $S/mlp_fused_split.hip 96  launch_mlp_fused_split
build mlp_c192_new $S/mlp_fused_wide.hip  192 launch_mlp_fused_wide
build mlp_c384_new $S/mlp_fused_wide.hip  384 launch_mlp_fused_wide
if [ -d $O/old ]; then
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DWIDE_C=96 -DWIDE_FN=launch_mlp_fused_split -DWIDE_SRC="\"$PWD/$O/old/mlp_fused_split.hip\"" tools/wide_lab.hip -o $O/mlp_c96_old &
  build mlp_c192_old $PWD/$O/old/mlp_fused_wide.hip 192 launch_mlp_fused_wide
  build mlp_c384_old $PWD/$O/old/mlp_fused_wide.hip 384 launch_mlp_fused_wide
fi
wait
ls -la $O
