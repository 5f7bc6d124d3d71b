#!/bin/bash
# Ablation variants of gemm_bf16_kernel (gemm_bf16.hip), built HERE into build/labs/ablg_*; run on the GPU box:
#   for b in build/labs/ablg_*; do echo "$(basename $b):"; $b; done
# Outputs of the variants are wrong by construction; only their run time matters.
set -e
cd "$(dirname "$0")/../.."
S=$PWD/audioset-convnext-inf_amd/csrc; O=build/labs; mkdir -p $O/src
variant() {   # name, sed expressions...
  local name=$1; shift
  local f=$O/src/ablg_$name.hip
  cp $S/gemm_bf16.hip $f
  for e in "$@"; do sed -i -E "$e" $f; done
  sed -i 's#"acx_internal.h"#"'$S'/acx_internal.h"#; s#"split_math.h"#"'$S'/split_math.h"#' $f
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DGEMM_SRC="\"$PWD/$f\"" tools/lab/gemm_bf16_lab.hip -o $O/ablg_$name &
}
NOEPI='s/^    if \(m0 \+ kBM <= p.M\) epilogue\(std::false_type\{\}\);/    if (p.M < 0) epilogue(std::false_type{});/; s/^    else epilogue\(std::true_type\{\}\);//'
NOBAR='s/^        __syncthreads\(\);$/        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");/'
NOREAD='s/^            af_\[i\] = \*reinterpret_cast<const f32x4\*>\(\(abase\) \+ i \* 32 \* kBfRowBytes \+ foff\[g\]\);        \\$/            af_[i] = __builtin_bit_cast(f32x4, uint4{(unsigned)(size_t)(abase), (unsigned)i, (unsigned)foff[g], 1u});        \\/; s/^            bf_\[j\] = \*reinterpret_cast<const f32x4\*>\(\(bbase\) \+ j \* 32 \* kBfRowBytes \+ foff\[g\]\);        \\$/            bf_[j] = __builtin_bit_cast(f32x4, uint4{(unsigned)(size_t)(bbase), (unsigned)j, (unsigned)foff[g], 2u});        \\/'
NODMA='s/^                lds_dma16_b\(src_\[i \* TN \+ j\] \+ \(koff_\), \(dst_\) \+ \(i \* TN \+ j\) \* 8 \* kBfRowBytes\);      \\$/                asm volatile("" :: "v"(src_[i * TN + j]));      \\/'
variant full
variant noepi "$NOEPI"
variant nobar "$NOBAR"
variant noread "$NOREAD"
variant nodma "$NODMA"
variant noepi_nodma "$NOEPI" "$NODMA"
variant noepi_noread "$NOEPI" "$NOREAD"
variant noepi_nodma_noread_nobar "$NOEPI" "$NODMA" "$NOREAD" "$NOBAR"
wait
for v in noepi nobar noread nodma; do echo "$v: $(diff $O/src/ablg_full.hip $O/src/ablg_$v.hip | grep -c '^[<>]') changed lines"; done
ls $O | grep ablg_
