#!/bin/bash
# Builds and times ablation variants of mlp_fused_wide.hip on the GPU box.  Each variant = sed script on the product source.
mkdir -p gpurun_out; O=gpurun_out/wide_lab.txt; : > $O
SRC=audioset-convnext-inf_amd/csrc/mlp_fused_wide.hip
variant() {   # name, sed expression(s)
  local name=$1; shift
  if [ -n "$ONLY" ] && [[ " $ONLY " != *" $name "* ]]; then return; fi
  cp $SRC /tmp/wide_variant.hip
  for e in "$@"; do sed -i -E "$e" /tmp/wide_variant.hip; done
  cp audioset-convnext-inf_amd/csrc/split_math.h /tmp/split_math_variant.h
  for e in "$@"; do sed -i -E "$e" /tmp/split_math_variant.h; done
  sed -i 's#"split_math.h"#"/tmp/split_math_variant.h"#' /tmp/wide_variant.hip
  sed -i 's#"acx_internal.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/acx_internal.h"#; s#"split_math.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/split_math.h"#' /tmp/wide_variant.hip
  if hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -w -DWIDE_SRC='"/tmp/wide_variant.hip"' tools/wide_lab.hip -o /tmp/wide_lab 2>>$O; then
    echo -n "$name: " >> $O; if [ -z "$DRY" ]; then /tmp/wide_lab >> $O; /tmp/wide_lab 32768 >> $O; else echo built >> $O; fi
  else echo "$name: BUILD FAILED" >> $O; fi
}
variant full
variant no_dma 's/^        __builtin_amdgcn_global_load_lds\(/        if (0) __builtin_amdgcn_global_load_lds(/'
variant no_gelu 's/if constexpr \(HV_\) \{ if \(\(\(m_\) . 1\) == 1\)/if constexpr (false) { if (((m_) \& 1) == 1)/'
variant gelu_p2_only 's/ACX_AFTER_MFMA\(HV, 1, /ACX_AFTER_MFMA(false, 1, /'
variant gelu_p1_only 's/ACX_AFTER_MFMA\(HV, 0, /ACX_AFTER_MFMA(false, 0, /'
variant gelu_no_trans 's/__builtin_amdgcn_rcpf\(([^)]*)\)/(\1 * 0.5f)/g; s/__builtin_amdgcn_exp2f\(([^)]*)\)/(\1 * 0.25f)/g'
variant no_barrier 's/__builtin_amdgcn_s_barrier\(\);/ /'
variant no_p1_mfma 's/^(\s+)X = __builtin_amdgcn_mfma_f32_32x32x16_f16\(.*$/\1asm volatile("" :: "v"(ah_), "v"(al_)); \\/'
variant no_ldsread 's/#define ACX_W1_RD\(base_, s_, pl_\).*/#define ACX_W1_RD(base_, s_, pl_) (acth[(s_) % 4])/; s/#define ACX_W2_RD\(base_, i_, pl_\).*/#define ACX_W2_RD(base_, i_, pl_) (actl[(i_) % 4])/'
variant no_mfma_fence 's/^        ACX_FENCE if constexpr \(HV_\) \{ (ACX_MICRO_RANGE.*) \} ACX_FENCE$/        if constexpr (HV_) { \1 }/'
variant no_fence_at_all 's/^#define ACX_FENCE __builtin_amdgcn_sched_barrier\(0\);/#define ACX_FENCE/'
cat $O
