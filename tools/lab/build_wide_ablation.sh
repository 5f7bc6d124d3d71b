#!/bin/bash
# Ablation variants of the wide fused MLP kernel (mlp_fused_wide.hip), built HERE into build/labs/ablw<C>_*; run on the GPU box:
#   for b in build/labs/ablw384_*; do echo -n "$(basename $b): "; $b; done
# Outputs of the variants are wrong by construction; only their run time matters.
set -e
cd "$(dirname "$0")/../.."
S=$PWD/audioset-convnext-inf_amd/csrc; O=build/labs; mkdir -p $O/src
C=${1:-384}
variant() {   # name, sed expressions...
  local name=$1; shift
  cp $S/mlp_fused_wide.hip $O/src/ablw${C}_$name.hip
  for e in "$@"; do sed -i -E "$e" $O/src/ablw${C}_$name.hip; done
  sed -i 's#"acx_internal.h"#"'$S'/acx_internal.h"#; s#"split_math.h"#"'$S'/split_math.h"#' $O/src/ablw${C}_$name.hip
  hipcc -O3 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 --offload-arch=gfx950 -w $EXTRA -DWIDE_C=$C -DWIDE_FN=launch_mlp_fused_wide -DWIDE_SRC="\"$PWD/$O/src/ablw${C}_$name.hip\"" tools/wide_lab.hip -o $O/ablw${C}_$name &
}
variant full
EXTRA=-DACX_FW_STAMPS variant stamps
variant nogelu 's/^(\s+)ACX_FENCE if constexpr \(HV_\) \{ ACX_NANO_RANGE.*$/\1ACX_FENCE/'
variant nodma 's/acx_glds16_run\(wbase, \(piece_\) % 8\);//'
variant nobarrier 's/^        __builtin_amdgcn_s_barrier\(\);  /  /'
variant nodsread 's/^#define ACX_W1_RD\(base_, u_, pl_\).*/#define ACX_W1_RD(base_, u_, pl_) (acth[0][(u_) % 4])/; s/^#define ACX_W2_RD\(base_, i_, pl_\).*/#define ACX_W2_RD(base_, i_, pl_) (actl[0][(i_) % 4])/'
variant nomfma 's/^#define ACX_M16\(a_, b_, c_\).*/#define ACX_M16(a_, b_, c_) asm volatile("" :: "v"(a_), "v"(b_));/'
EXTRA=-DACX_FW_STAMPS variant stamps_nomfma 's/^#define ACX_M16\(a_, b_, c_\).*/#define ACX_M16(a_, b_, c_) asm volatile("" :: "v"(a_), "v"(b_));/'
NOGELU='s/^(\s+)ACX_FENCE if constexpr \(HV_\) \{ ACX_NANO_RANGE.*$/\1ACX_FENCE/; s/^        if constexpr \(HV\) \{ ACX_NANO_RANGE\([01], 0, Cfg::kNanoHead\) \}$/ /'
NODS='s/^#define ACX_W1_RD\(base_, u_, pl_\).*/#define ACX_W1_RD(base_, u_, pl_) (acth[0][(u_) % 4])/; s/^#define ACX_W2_RD\(base_, i_, pl_\).*/#define ACX_W2_RD(base_, i_, pl_) (actl[0][(i_) % 4])/'
NODMA='s/acx_glds16_run\(wbase, \(piece_\) % 8\);//'
EXTRA=-DACX_FW_STAMPS variant st_nogelu "$NOGELU"
EXTRA=-DACX_FW_STAMPS variant st_nodsread "$NODS"
EXTRA=-DACX_FW_STAMPS variant st_nodma "$NODMA"
EXTRA=-DACX_FW_STAMPS variant st_nogelu_nodsread "$NOGELU" "$NODS"
EXTRA=-DACX_FW_STAMPS variant st_nogelu_nodsread_nodma "$NOGELU" "$NODS" "$NODMA"
wait
for v in nogelu nodma nobarrier nodsread nomfma; do echo "$v: $(diff $O/src/ablw${C}_full.hip $O/src/ablw${C}_$v.hip | grep -c '^[<>]') changed lines"; done
ls $O | grep ablw$C
