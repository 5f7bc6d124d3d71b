#!/bin/bash
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session4.txt
: > $O
run() { echo "### ACX_LIB=$(basename ${ACX_LIB:-libacx.so}) $*" >> $O; timeout 300 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
run python tools/race2/run_classes.py down2
run python tools/race2/run_classes.py block2
run python tools/race2/run_probe.py down2 2
cat $O
