#!/bin/bash
# A/B of libacx builds on ONE box, interleaved twice: bench.py with each of the named libraries in turn.
#   tools/lab/ab_bench.sh new r02 wideR02 ...     ("new" = the in-tree library; <name> = build/labs/libacx_<name>.so)
L=audioset-convnext-inf_amd/libacx.so
cp $L /tmp/libacx_new.so
show() { python - "$1" <<'EOF2'
import json, sys
d = json.load(open(sys.argv[1]))
print("%8.1f clips/s %6.3f ms | " % (d["value"], d["ms_per_step"]) + " ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
EOF2
}
for rep in 1 2; do
  for which in "$@"; do
    if [ $which = new ]; then cp /tmp/libacx_new.so $L; else cp build/labs/libacx_$which.so $L; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > /tmp/b_$which.json 2>/tmp/b_$which.err || tail -3 /tmp/b_$which.err
    printf "%-10s" "$which:"; show /tmp/b_$which.json
  done
done
cp /tmp/libacx_new.so $L
