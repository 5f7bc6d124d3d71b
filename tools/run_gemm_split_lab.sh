#!/bin/bash
# Builds and times ablation variants of gemm_split.hip on the GPU box.  Each variant = sed script on the product source.
mkdir -p gpurun_out; O=gpurun_out/gemm_split_lab.txt; : > $O
SRC=audioset-convnext-inf_amd/csrc/gemm_split.hip
variant() {   # name, sed expression(s)
  local name=$1; shift
  cp $SRC /tmp/gemm_variant.hip
  for e in "$@"; do sed -i -E "$e" /tmp/gemm_variant.hip; done
  sed -i 's#"acx_internal.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/acx_internal.h"#; s#"split_math.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/split_math.h"#' /tmp/gemm_variant.hip
  if hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DGEMM_SRC='"/tmp/gemm_variant.hip"' tools/gemm_split_lab.hip -o /tmp/gemm_split_lab 2>>$O; then
    echo "== $name" >> $O; if [ -z "$DRY" ]; then /tmp/gemm_split_lab >> $O; else echo built >> $O; fi
  else echo "$name: BUILD FAILED" >> $O; fi
}
variant full
variant no_loop_dma 's/^            if \(pc < B_DMA\) lds_dma16_s/            if (0) lds_dma16_s/; s/^                lds_dma16_s\(a_src\[pc - B_DMA\]/                if (0) lds_dma16_s(a_src[pc - B_DMA]/'
variant no_barrier 's/^        __builtin_amdgcn_s_barrier\(\);/ /'
variant no_fragreads 's/^        ACX_READ_FRAGS\(F0, abn, bbn, 0\)/ /; s/^        ACX_READ_FRAGS\(F1, abn, bbn, 1\)/ /'
variant no_mfma_dma_fence 's/^            __builtin_amdgcn_sched_barrier\(0\);                                                         \\$/ \\/'
cat $O
