#!/bin/bash
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session3.txt
: > $O
run() { echo "### ACX_LIB=$(basename ${ACX_LIB:-libacx.so}) $*" >> $O; timeout 300 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
run python tools/race2/run_classes.py none
run python tools/race2/run_classes.py down2
run python tools/race2/run_classes.py block2
run python tools/race2/run_classes.py burn4 256
run python tools/race2/run_classes.py burn4 1024
run python tools/race2/run_classes.py burn4 2048
run python tools/race2/run_classes.py burn1 1024
run python tools/race2/run_classes.py burn6 1024
ACX_PRECISION=fp32 run python tools/race2/run_classes.py block2
ACX_LIB=$PWD/build/variants/libacx_oneterm.so run python tools/race2/run_classes.py down2
cat $O
