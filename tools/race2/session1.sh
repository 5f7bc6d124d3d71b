#!/bin/bash
# GPU session 1: which kernels are aggressors, what kind of corruption, is register staging clean?
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session1.txt
: > $O
run() { echo "### $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
run python tools/race2/run_probe.py none 3
run python tools/race2/run_probe.py down2 8
run python tools/race2/run_probe.py block2 6
run python tools/race2/run_probe.py block0 6
run python tools/race2/run_probe.py block1 6
run python tools/race2/run_probe.py block3 6
ACX_LIB=$PWD/build/variants/libacx_syncstage.so run python tools/race2/run_probe.py down2 8
ACX_LIB=$PWD/build/variants/libacx_syncstage.so run python tools/race2/run_probe.py block2 6
ACX_PRECISION=fp32 run python tools/race2/run_probe.py block2 4
ACX_PRECISION=bf16 run python tools/race2/run_probe.py block2 4
tail -n 400 $O
