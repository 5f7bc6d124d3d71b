#!/bin/bash
ACX_RECORD_FLOOR=$PWD/gpurun_out/floor_new.json python -m pytest tests/test_gpu_frontend_edge.py -q -k "dense_frontend_is" 2>&1 | tail -2
python tests/parity_floor.py merge gpurun_out/floor_new.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== A/B: dwconv7_mfma exclusive (new) vs two workgroups per CU (base)"
bash tools/lab/ab_lib.sh build/labs/libacx_dwold.so bf16a
echo "== default bench line"
python bench.py --steps 20 --warmup 5 > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; tail -2 gpurun_out/b_bench.err
python - <<'EOF2'
import json
d=json.load(open("gpurun_out/b_bench.json"))
print("headline %.0f clips/s %.3f ms; frac %.3f; bf16a %.0f; frame256 %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["bf16a_shard"]["value"], d["frame_bs256"]["value"]))
print("eval_sweep", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("eval_sweep", {}).items() if k not in ("workload",)})
EOF2
