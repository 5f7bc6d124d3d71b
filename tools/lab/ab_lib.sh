#!/bin/bash
# On the GPU box: same-box A/B of two builds of libacx.so through the ACX_LIB switch (audioset-convnext-inf_amd/_ffi.py).
#   bash tools/lab/ab_lib.sh build/labs/libacx_head.so [precision ...]       alternates base / new, three rounds each
BASE=$1; shift
PRECS=${@:-fp32_split bf16a}
summ() { python - "$1" <<'EOF2'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print("%.0f clips/s %.3f ms | " % (d["value"], d["ms_per_step"]) + " ".join("%s %.3f" % (n, k[n]["ms_per_step"]) for n in k))
EOF2
}
for P in $PRECS; do
  for i in 1 2 3; do
    for which in base new; do
      if [ $which = base ]; then export ACX_LIB=$PWD/$BASE; else unset ACX_LIB; fi
      python bench.py --precision $P --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > /tmp/ab.json 2>/dev/null && echo -n "$P $which: " && summ /tmp/ab.json
    done
  done
done
