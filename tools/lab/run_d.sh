#!/bin/bash
# full GPU suite, then the default bench line (live traffic + eval-sweep child)
SECONDS=0; python bench.py > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err; echo "bench.py wall: $SECONDS s"; grep -E "bench.py:" gpurun_out/d_bench.err
python - <<'EOF2'
import json
d=json.load(open("gpurun_out/d_bench.json"))
r=d["roofline"]; w=d["roofline_dwconv"]
print("headline %.0f clips/s %.3f ms steps %d; frac %.3f traffic %s src %s clock %s" % (d["value"], d["ms_per_step"], d["steps"], r["frac"], r["traffic"], r["traffic_source"], r.get("shader_clock_GHz_in_pmc_pass")))
print("dwconv frac %.3f traffic %s alg %s" % (w["frac"], w["traffic"], w["algorithmic_bytes_per_launch"]))
print("bf16a %.0f frac %.3f dw %.3f; frame256 %.0f" % (d["bf16a_shard"]["value"], d["bf16a_shard"]["roofline"]["frac"], d["bf16a_shard"]["roofline_dwconv"]["frac"], d["frame_bs256"]["value"]))
print("eval_sweep", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("eval_sweep", {}).items() if k not in ("workload",)})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
EOF2
