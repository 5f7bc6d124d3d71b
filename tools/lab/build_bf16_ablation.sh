#!/bin/bash
# Ablation variants of the bf16 fused MLP kernels (mlp_fused_wide_bf16.hip), built HERE into build/labs/ablb<C>_<abf>_*; run on the GPU box:
#   for b in build/labs/ablb*; do echo -n "$(basename $b): "; $b; done
# usage: build_bf16_ablation.sh <C: 96|192|384> <activations in HBM: 0 = fp32, 1 = bf16>
# Outputs of the variants are wrong by construction; only their run time matters.
set -e
cd "$(dirname "$0")/../.."
S=$PWD/audioset-convnext-inf_amd/csrc; O=build/labs; mkdir -p $O/src
C=${1:-96}; ABF=${2:-0}
variant() {   # name, sed expressions...
  local name=$1; shift
  local f=$O/src/ablb${C}_${ABF}_$name.hip
  cp $S/mlp_fused_wide_bf16.hip $f
  for e in "$@"; do sed -i -E "$e" $f; done
  sed -i 's#"acx_internal.h"#"'$S'/acx_internal.h"#; s#"split_math.h"#"'$S'/split_math.h"#' $f
  hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -w $EXTRA -DWIDE_C=$C -DWIDE_BF16=$ABF -DWIDE_SRC="\"$PWD/$f\"" tools/wide_lab.hip -o $O/ablb${C}_${ABF}_$name &
}
NOGELU='s/gelu3_micro<[0-9]>\(gs, gk, ax_?, ay_?\);?/;/g; s/pack_bf16\(gs\.qx, gs\.qy\)/pack_bf16(ax_, ay_)/'
NOMFMA='s/__builtin_amdgcn_mfma_f32_32x32x16_bf16\(/acx_fake_mfma(/; s/^namespace acx \{$/namespace acx { typedef float f32x16_ __attribute__((ext_vector_type(16))); template <class A, class B> __device__ __forceinline__ f32x16_ acx_fake_mfma(A a, B b, f32x16_ c, int, int, int) { asm volatile("" :: "v"(a), "v"(b)); return c; }/'
NODSREAD='s/^#define ACX_W1_RD\(base_, u_\).*/#define ACX_W1_RD(base_, u_) (ACX_ACT0[(u_) % 4])/; s/^#define ACX_W2_RD\(base_, u_\).*/#define ACX_W2_RD(base_, u_) (ACX_ACT0[((u_) + 1) % 4])/'
variant full
variant nogelu "$NOGELU"
variant nomfma "$NOMFMA"
variant nogelu_nomfma "$NOGELU" "$NOMFMA"
wait
ls $O | grep ablb${C}_${ABF}
