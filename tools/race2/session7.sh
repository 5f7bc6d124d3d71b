#!/bin/bash
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session7.txt
: > $O
run() { echo "### DET_LDS=$DET_LDS ACX_LIB=$(basename ${ACX_LIB:-libacx.so}) $*" >> $O; timeout 600 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
ACX_LIB=$PWD/build/variants/libacx_excl.so run python tools/race2/run_detect.py down2 down3 block2 block3
ACX_LIB=$PWD/build/variants/libacx_excl.so run python tools/race2/run_probe.py down2 4
ACX_LIB=$PWD/build/variants/libacx_excl.so run python tools/race_torchvictim.py
run python tools/race2/run_detect.py down2
cat $O
