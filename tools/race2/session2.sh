#!/bin/bash
# GPU session 2: LDS reads vs VALU, MFMA density, CU-local vs chip-wide
mkdir -p gpurun_out/race2
O=gpurun_out/race2/session2.txt
: > $O
run() { echo "### ACX_LIB=$(basename ${ACX_LIB:-libacx.so}) $*" >> $O; timeout 300 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo "rc=$?" >> $O; }
run python tools/race2/run_probe.py down2 6
run python tools/race2/run_probe.py block2 4
for v in oneterm lds80 lds120; do
  ACX_LIB=$PWD/build/variants/libacx_$v.so run python tools/race2/run_probe.py down2 6
done
ACX_LIB=$PWD/build/variants/libacx_lds120.so run python tools/race2/run_probe.py block2 4
tail -n 300 $O
