#!/bin/bash
# On the GPU box: the headline forward under a few runtime knobs (alternating, two rounds)
hl() { env "$@" python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-profile $PREC 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   %-50s %.0f clips/s %.3f ms' % ('$*', d['value'], d['ms_per_step']))"; }
for PREC in "" "--precision bf16a"; do
echo "== $PREC"
for r in 1 2; do
hl A=1
hl HIP_FORCE_DEV_KERNARG=1
hl GPU_MAX_HW_QUEUES=2 ACX_SPLIT_WAYS=3
hl GPU_MAX_HW_QUEUES=3 ACX_SPLIT_WAYS=3
hl HSA_NO_SCRATCH_RECLAIM=1
done; done
