#!/bin/bash
# Builds and times ablation variants of mlp_fused_wide_bf16.hip on the GPU box.  Each variant = sed script on the product source.
# usage: tools/run_wide_bf16_lab.sh [C ...]   (default 96 192 384)
mkdir -p gpurun_out; O=gpurun_out/wide_bf16_lab.txt; : > $O
SRC=${SRC:-audioset-convnext-inf_amd/csrc/mlp_fused_wide_bf16.hip}
CS=${@:-96 192 384}
variant() {   # name, sed expression(s)
  local name=$1; shift
  if [ -n "$ONLY" ] && [[ " $ONLY " != *" $name "* ]]; then return; fi
  cp $SRC /tmp/wide_variant.hip
  for e in "$@"; do sed -i -E "$e" /tmp/wide_variant.hip; done
  cp audioset-convnext-inf_amd/csrc/split_math.h /tmp/split_math_variant.h
  for e in "$@"; do sed -i -E "$e" /tmp/split_math_variant.h; done
  sed -i 's#"split_math.h"#"/tmp/split_math_variant.h"#' /tmp/wide_variant.hip
  sed -i 's#"acx_internal.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/acx_internal.h"#' /tmp/wide_variant.hip
  for C in $CS; do
    if hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -w -DWIDE_C=$C -DWIDE_SRC='"/tmp/wide_variant.hip"' tools/wide_bf16_lab.hip -o /tmp/wide_bf16_lab 2>>$O; then
      echo -n "$name: " >> $O; if [ -z "$DRY" ]; then /tmp/wide_bf16_lab >> $O; else echo built >> $O; fi
    else echo "$name C=$C: BUILD FAILED" >> $O; fi
  done
}
variant full
variant no_dma 's/^        __builtin_amdgcn_global_load_lds\(/        if (0) __builtin_amdgcn_global_load_lds(/'
variant no_gelu 's/gelu_micro<0>\(gs, gk, dummy_, dummy_\); \}/gs.gx = gs.ax; gs.gy = gs.ay; }/; s/else if \(st_ == ([1-6])\) gelu_micro<[1-6]>\(gs, gk, dummy_, dummy_\);/else if (st_ == \1) {}/'
variant gelu_no_trans 's/__builtin_amdgcn_rcpf\(([^)]*)\)/(\1 * 0.5f)/g; s/__builtin_amdgcn_exp2f\(([^)]*)\)/(\1 * 0.25f)/g'
variant no_barrier 's/__builtin_amdgcn_s_barrier\(\);/ /'
variant one_chunk_pair 's/for \(int k = 1; k < n - 1; \+\+k\) \{/for (int k = 1; k < 2; ++k) {/'
variant no_ldsread 's/#define ACX_W1_RD\(base_, u_\).*/#define ACX_W1_RD(base_, u_) (act[0][(u_) % 4])/; s/#define ACX_W2_RD\(base_, u_\).*/#define ACX_W2_RD(base_, u_) (act[0][(u_) % 4])/'
variant ring_hot_rows 's/mrow\[pt\] = \(tile_\) \* Cfg::kPix \+/mrow[pt] = ((tile_) \& 1) * Cfg::kPix +/'
# the weights-stationary C = 96 kernel
variant stat_no_gelu 's/gelu_micro<0>\(gs, gk, dummy, dummy\); gelu_micro<1>\(gs, gk, dummy, dummy\); gelu_micro<2>\(gs, gk, dummy, dummy\);/gs.gx = gs.ax; gs.gy = gs.ay;/; s/gelu_micro<[3-6]>\(gs, gk, dummy, dummy\);//g'
variant stat_one_chunk 's/for \(int k = 0; k < n; \+\+k\) \{/for (int k = 0; k < 1; ++k) {/'
variant stat_no_chunks 's/for \(int k = 0; k < n; \+\+k\) \{/for (int k = 0; k < (int)(M >> 40); ++k) {/'
variant stat_no_mfma 's/X\[u \& 1\] = __builtin_amdgcn_mfma_f32_32x32x16_bf16\(ACX_B8\(f\), ACX_B8\(act\[u >> 1\]\), X\[u \& 1\], 0, 0, 0\);/X[u \& 1][u] += f[0] * act[u >> 1][1];/; s/acc\[u >> 2\] = __builtin_amdgcn_mfma_f32_32x32x16_bf16\(ACX_B8\(f\), ACX_B8\(g\[u \& 3\]\), acc\[u >> 2\], 0, 0, 0\);/acc[u >> 2][u] += f[0] * g[u \& 3][1];/'
variant stat_hot_rows 's/long long mrow = tile \* 32 \+ l31;/long long mrow = (tile \& 1) * 32 + l31;/; s/long long r_ = \(tile_\) \* 32 \+ l31;/long long r_ = ((tile_) \& 1) * 32 + l31;/'
variant stat_hot_rows_no_gelu 's/long long mrow = tile \* 32 \+ l31;/long long mrow = (tile \& 1) * 32 + l31;/; s/long long r_ = \(tile_\) \* 32 \+ l31;/long long r_ = ((tile_) \& 1) * 32 + l31;/' 's/gelu_micro<0>\(gs, gk, dummy, dummy\); gelu_micro<1>\(gs, gk, dummy, dummy\); gelu_micro<2>\(gs, gk, dummy, dummy\);/gs.gx = gs.ax; gs.gy = gs.ay;/; s/gelu_micro<[3-6]>\(gs, gk, dummy, dummy\);//g'
cat $O
