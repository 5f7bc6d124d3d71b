#!/bin/bash
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session6.txt
: > $O
run() { echo "### DET_LDS=$DET_LDS ACX_LIB=$(basename ${ACX_LIB:-libacx.so}) $*" >> $O; timeout 600 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
run python tools/race2/run_detect.py down2 down2z burn8:256 burn8:1024 burn8:2048
DET_LDS=163840 run python tools/race2/run_detect.py down2 block3 block0
DET_LDS=65536 run python tools/race2/run_detect.py down2
cat $O
