#!/bin/bash
# Ablation variants of the stage-0 fused MLP kernel (mlp_fused_split.hip), built HERE into build/labs/abl96_*; run on the GPU box:
#   for b in build/labs/abl96_*; do echo -n "$(basename $b): "; $b; done
# Outputs of the variants are wrong by construction; only their run time matters.
set -e
cd "$(dirname "$0")/../.."
S=$PWD/audioset-convnext-inf_amd/csrc; O=build/labs; mkdir -p $O/src
variant() {   # name, sed expressions...
  local name=$1; shift
  cp $S/mlp_fused_split.hip $O/src/abl96_$name.hip
  for e in "$@"; do sed -i -E "$e" $O/src/abl96_$name.hip; done
  sed -i 's#"acx_internal.h"#"'$S'/acx_internal.h"#; s#"split_math.h"#"'$S'/split_math.h"#' $O/src/abl96_$name.hip
  hipcc -O3 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 --offload-arch=gfx950 -w $EXTRA -DWIDE_C=96 -DWIDE_FN=launch_mlp_fused_split -DWIDE_SRC="\"$PWD/$O/src/abl96_$name.hip\"" tools/wide_lab.hip -o $O/abl96_$name &
}
variant full
EXTRA=-DACX_FS_STAMPS variant stamps
EXTRA=-DACX_FS_STAMPS variant stamps_nogelu 's/^#define ACX_AFTER\(m_\).*/#define ACX_AFTER(m_) ACX_FENCE/'
EXTRA=-DACX_FS_STAMPS variant stamps_nomfma 's/^(\s+)(Xacc|acc\[\(i_\) >> 1\]) = __builtin_amdgcn_mfma_f32_32x32x16_f16\((ACX_H8\([a-z_]+\)), (ACX_H8\([^)]*\)\)?), .*$/\1asm volatile("" :: "v"(\3), "v"(\4)); \\/'
EXTRA=-DACX_FS_STAMPS variant stamps
EXTRA=-DACX_FS_STAMPS variant stamps_nogelu 's/^#define ACX_AFTER\(m_\).*/#define ACX_AFTER(m_) ACX_FENCE/'
EXTRA=-DACX_FS_STAMPS variant stamps_nomfma 's/^(\s+)(Xacc|acc\[\(i_\) >> 1\]) = __builtin_amdgcn_mfma_f32_32x32x16_f16\((ACX_H8\([a-z_]+\)), (ACX_H8\([^)]*\)\)?), .*$/\1asm volatile("" :: "v"(\3), "v"(\4)); \\/'
variant nogelu 's/^#define ACX_AFTER\(m_\).*/#define ACX_AFTER(m_) ACX_FENCE/'
variant nodma 's/^#define ACX_DMA_AT\(region_\).*/#define ACX_DMA_AT(region_) \\/'
variant nobarrier 's/^            __builtin_amdgcn_s_barrier\(\);$/ /'
variant nowait 's/^            if \(role == 0 \&\& wdma \&\& !LAST\) asm volatile.*$/ /; s/^            else asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\);$/ /'
variant nodma_nowait_nobar 's/^#define ACX_DMA_AT\(region_\).*/#define ACX_DMA_AT(region_) \\/; s/^            __builtin_amdgcn_s_barrier\(\);$/ /; s/^            if \(role == 0 \&\& wdma \&\& !LAST\) asm volatile.*$/ /; s/^            else asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\);$/ /'
variant nodsread 's/^#define ACX_W1_RD\(w1p_, s_, pl_\).*/#define ACX_W1_RD(w1p_, s_, pl_) (acth[(s_) % 4])/; s/^#define ACX_W2_RD\(w2p_, i_, pl_\).*/#define ACX_W2_RD(w2p_, i_, pl_) (actl[(i_) % 4])/'
variant nomfma 's/^(\s+)(Xacc|acc\[\(i_\) >> 1\]) = __builtin_amdgcn_mfma_f32_32x32x16_f16\((ACX_H8\([a-z_]+\)), (ACX_H8\([^)]*\)\)?), .*$/\1asm volatile("" :: "v"(\3), "v"(\4)); \\/'
wait
ls $O | grep abl96
