#!/bin/bash
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session5.txt
: > $O
run() { echo "### ACX_LIB=$(basename ${ACX_LIB:-libacx.so}) ACX_PRECISION=$ACX_PRECISION $*" >> $O; timeout 600 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
run python tools/race2/run_detect.py none down2 down1 down3 block0 block1 block2 block3 dw0 dw2 burn1:1024 burn4:256 burn4:1024 burn4:2048 burn6:1024
ACX_PRECISION=fp32 run python tools/race2/run_detect.py block0 block2 down2
ACX_PRECISION=bf16 run python tools/race2/run_detect.py block2 down2
for v in oneterm syncstage lds120; do ACX_LIB=$PWD/build/variants/libacx_$v.so run python tools/race2/run_detect.py down2 block2; done
cat $O
