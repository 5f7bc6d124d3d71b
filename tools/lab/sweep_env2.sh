#!/bin/bash
# On the GPU box: the evaluation sweep (bench.py --eval-sweep-only, a process of its own) under a few runtime settings
run() { echo "== $*"; env "$@" python bench.py --eval-sweep-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('ACX_EVAL_SWEEP '):
        d = json.loads(l[15:]); print('   sweep %.0f steady %.0f resident %.0f  -> %.3f / %.3f' % (d['value'], d['steady_state_clips_per_s'], d['resident_bs256_clips_per_s'], d['vs_resident_bs256'], d['steady_state_vs_resident_bs256']))"; }
run A=1
run A=1
run ACX_SPLIT_STREAMS=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_SDMA=0
