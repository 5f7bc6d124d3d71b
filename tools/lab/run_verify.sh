#!/bin/bash
# final verification of the tree on a GPU box: smoke(), the full GPU suite, the default bench line
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
SECONDS=0; python bench.py > gpurun_out/v_bench.json 2> gpurun_out/v_bench.err; echo "bench.py wall: $SECONDS s"
python - <<'EOF2'
import json
d=json.load(open("gpurun_out/v_bench.json"))
r=d["roofline"]; w=d["roofline_dwconv"]
print("headline %.0f clips/s %.3f ms; frac %.3f traffic %.0f MB (%s); dw %.3f; queues %s" % (d["value"], d["ms_per_step"], r["frac"], r["traffic"]/1e6, r["traffic_source"][:30], w["frac"], d["config"]["gpu_max_hw_queues"]))
print("bf16a %.0f frac %.3f dw %.3f; frame256 %.0f; native %.0f" % (d["bf16a_shard"]["value"], d["bf16a_shard"]["roofline"]["frac"], d["bf16a_shard"]["roofline_dwconv"]["frac"], d["frame_bs256"]["value"], d["native_f32_mfma"]["value"]))
e=d["eval_sweep"]; print("eval_sweep %.0f, %.3f / %.3f of resident" % (e["value"], e["vs_resident_bs256"], e["steady_state_vs_resident_bs256"]))
print("cpu %.1f clips/s on %d cores -> %.0fx" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["value"]/d["cpu_baseline"]["value"]))
EOF2
