#!/bin/bash
# Builds and times ablation variants of mlp_fused_widep.hip (C = 96) on the GPU box.
mkdir -p gpurun_out; O=gpurun_out/widep_lab.txt; : > $O
SRC=tools/experimental/mlp_fused_widep.hip
variant() {   # name, sed expression(s)
  local name=$1; shift
  cp $SRC /tmp/wide_variant.hip
  cp audioset-convnext-inf_amd/csrc/split_math.h /tmp/split_math_variant.h
  for e in "$@"; do sed -i -E "$e" /tmp/wide_variant.hip; sed -i -E "$e" /tmp/split_math_variant.h; done
  sed -i 's#"split_math.h"#"/tmp/split_math_variant.h"#' /tmp/wide_variant.hip
  sed -i 's#"acx_internal.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/acx_internal.h"#' /tmp/wide_variant.hip
  if hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -fno-slp-vectorize -DWIDE_C=96 -DWIDE_FN=launch_mlp_fused_widep -DWIDE_SRC='"/tmp/wide_variant.hip"' tools/wide_lab.hip -o /tmp/wide_lab 2>>$O; then
    echo -n "$name: " >> $O; if [ -z "$DRY" ]; then /tmp/wide_lab >> $O; else echo built >> $O; fi
  else echo "$name: BUILD FAILED" >> $O; fi
}
variant full
variant no_dma 's/^        __builtin_amdgcn_global_load_lds\(/        if (0) __builtin_amdgcn_global_load_lds(/'
variant no_gelu_steps 's/\{ ACX_MICRO_RANGE\(36 \* \(half_\).*\} \} ACX_FENCE/{ } } ACX_FENCE/'
variant no_barrier 's/__builtin_amdgcn_s_barrier\(\);/ /'
variant no_ldsread 's/#define ACX_W1_RD\(base_, s_, pl_\).*/#define ACX_W1_RD(base_, s_, pl_) (acth[(s_) % 4])/; s/#define ACX_W2_RD\(base_, i_, pl_\).*/#define ACX_W2_RD(base_, i_, pl_) (actl[(i_) % 4])/'
variant no_prefetch_wait 's/asm volatile\("s_waitcnt vmcnt\(%0\)" :: "n"\(W \+ \(e_\) <= 63 \? W \+ \(e_\) : 0\) : "memory"\);/asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W) : "memory");/'
variant no_io 's/#define ACX_ALOAD\(dst_, ptr_, offset_\).*/#define ACX_ALOAD(dst_, ptr_, offset_) asm volatile("v_mov_b32 %0, 0" : "=v"(dst_[0]));/; s/\*reinterpret_cast<f32x4\*>\(xp \+ c\) = v;/if (v[0] == 12345.678f) *reinterpret_cast<f32x4*>(xp + c) = v;/'
cat $O
