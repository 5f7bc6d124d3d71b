#!/bin/bash
# usage: flagsweep.sh  (on the GPU box) -- stage-0 split kernel and GEMM with a few scheduler flags
cd $GRAFT_REPO_ROOT
O=gpurun_out/flagsweep.txt; : > $O
SRC1=audioset-convnext-inf_amd/csrc/mlp_fused_split.hip
SRC2=audioset-convnext-inf_amd/csrc/gemm_split.hip
prep() { cp $1 /tmp/v.hip; sed -i 's#"acx_internal.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/acx_internal.h"#; s#"split_math.h"#"'$PWD'/audioset-convnext-inf_amd/csrc/split_math.h"#' /tmp/v.hip; }
for F in "" "-mllvm -enable-post-misched=0" "-mllvm -enable-misched=0" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-O2" "-mllvm -amdgpu-use-amdgpu-trackers=1"; do
  echo "== flags: [$F]" >> $O
  prep $SRC1
  if hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $F -DWIDE_C=96 -DWIDE_FN=launch_mlp_fused_split -DWIDE_SRC='"/tmp/v.hip"' tools/wide_lab.hip -o /tmp/l1 2>>$O; then /tmp/l1 >> $O; else echo "stage0: build failed" >> $O; fi
  prep $SRC2
  if hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $F -DGEMM_SRC='"/tmp/v.hip"' tools/gemm_split_lab.hip -o /tmp/l2 2>>$O; then /tmp/l2 | head -3 >> $O; else echo "gemm: build failed" >> $O; fi
done
cat $O
