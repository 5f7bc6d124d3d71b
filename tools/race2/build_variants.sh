#!/bin/bash
# ROUND-2 RECORD: builds the diagnostic variants of libacx that sessions 1-8 ran.  They were compiled from
# kernel sources WITH lab switches (tools/lab_src/ at commit eaa811c, removed from the tree in round 3 because the copies had
# drifted from csrc/): check that commit out to rebuild them.  The round-3 way: tools/lab/build_variant_lib.sh + an override
# directory (e.g. a gemm_split.hip without the CU-exclusive claim), tools/race2/run_detect.py with ACX_LIB=...,
# and tools/race2/standalone_repro.hip (no libacx at all).
set -e
cd "$(dirname "$0")/../.."
CS=audioset-convnext-inf_amd/csrc
mkdir -p build/variants
build() {   # name, file to recompile, flags
  local name=$1 file=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fvisibility=hidden -DACX_BUILD "$@" -c tools/lab_src/$file.hip -o build/variants/$name.$file.o
  local objs=""
  for f in api frontend stem dwconv gemm gemm_bf16 gemm_split mlp_fused mlp_fused_split misc; do
    if [ "$f" = "$file" ]; then objs="$objs build/variants/$name.$file.o"; else objs="$objs build/acx/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libacx_$name.so $objs
}
build syncstage gemm_split -DACX_DBG_SYNC_STAGE &
build oneterm gemm_split -DACX_DBG_ONE_TERM &
build lds80 gemm_split -DACX_DBG_LDS80 &
build lds120 gemm_split -DACX_DBG_LDS120 &
build excl gemm_split -DACX_DBG_EXCL &
wait
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -shared -fPIC -o build/variants/libfeprobe.so tools/race2/fe_probe.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -shared -fPIC -o build/variants/libvalucls.so tools/race2/valu_classes.hip
ls -la build/variants/*.so
