#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --eval-sweep-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('ACX_EVAL_SWEEP '):
        d = json.loads(l[15:]); print('   sweep %.0f steady %.0f resident %.0f  -> %.3f / %.3f' % (d['value'], d['steady_state_clips_per_s'], d['resident_bs256_clips_per_s'], d['vs_resident_bs256'], d['steady_state_vs_resident_bs256']))"; }
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=3
run GPU_MAX_HW_QUEUES=2
run A=1
hl() { echo "== headline $*"; env "$@" python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   %.0f clips/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
hl A=1
hl GPU_MAX_HW_QUEUES=2
hl A=1
hl GPU_MAX_HW_QUEUES=2
for p in bf16a; do
echo "== bf16a"; for e in A=1 GPU_MAX_HW_QUEUES=2 A=1 GPU_MAX_HW_QUEUES=2; do env $e python bench.py --precision bf16a --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   $e %.0f clips/s %.3f ms' % (d['value'], d['ms_per_step']))"; done; done
