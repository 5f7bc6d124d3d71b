#!/bin/bash
# On the GPU box: headline bench against the number of hardware queues HIP multiplexes streams onto (GPU_MAX_HW_QUEUES) and with the
# sub-batch split off -- does the mapping of the library's fork / join streams onto hardware queues matter?
for rep in 1 2; do
for q in 1 2 4 8; do
  echo -n "GPU_MAX_HW_QUEUES=$q: "; GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-profile | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
done
echo -n "default queues, ACX_SPLIT_STREAMS=0: "; ACX_SPLIT_STREAMS=0 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-profile | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
echo -n "default: "; timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-profile | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
done
