#!/bin/bash
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session8.txt
: > $O
run() { echo "### $*" >> $O; timeout 900 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
run python -m pytest tests -m gpu -x -q
run python tools/race2/run_detect.py down1 down2 down3 block0 block1 block2 block3
run python tools/race2/run_probe.py block2 3
run python tools/race_torchvictim.py
run python bench.py --steps 30 --warmup 5
ACX_SPLIT_WAYS=2 run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile
cat $O | cut -c1-2500
