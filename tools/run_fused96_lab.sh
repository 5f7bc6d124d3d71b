#!/bin/bash
# Ablation variants of mlp_fused_split.hip (stage 0, C = 96) on the GPU box: how much of a launch is the tile's memory phase?
mkdir -p gpurun_out; O=gpurun_out/fused96_lab.txt; : > $O
SRC=audioset-convnext-inf_amd/csrc/mlp_fused_split.hip
variant() {   # name, sed expression(s)
  local name=$1; shift
  cp $SRC /tmp/wide_variant.hip
  for e in "$@"; do sed -i -E "$e" /tmp/wide_variant.hip; done
  sed -i 's#"acx_internal.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/acx_internal.h"#; s#"split_math.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/split_math.h"#' /tmp/wide_variant.hip
  if hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DWIDE_C=96 -DWIDE_FN=launch_mlp_fused_split -DWIDE_SRC='"/tmp/wide_variant.hip"' tools/wide_lab.hip -o /tmp/fused96_lab 2>>$O; then
    echo -n "$name: " >> $O; if [ -z "$DRY" ]; then /tmp/fused96_lab >> $O; else echo built >> $O; fi
  else echo "$name: BUILD FAILED" >> $O; fi
}
variant full
variant hot_rows 's/const long long pix0 = \(long long\)blockIdx.x \* Cfg::kPix \+ wave \* 32;/const long long pix0 = (long long)(blockIdx.x \& 1) * Cfg::kPix + wave * 32;/'
cat $O
