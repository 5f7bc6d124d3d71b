#!/bin/bash
# final profile set of the round: bench line + rocprofv3 kernel stats (two streams / one stream) + PMC tables, both arithmetics
export GPU_MAX_HW_QUEUES=2
bash tools/collect_profiles.sh fp32_split > gpurun_out/collect_split.log 2>&1; tail -3 gpurun_out/collect_split.log | cut -c1-300
bash tools/collect_profiles.sh bf16a > gpurun_out/collect_bf16a.log 2>&1; tail -3 gpurun_out/collect_bf16a.log | cut -c1-300
python tools/latency.py > gpurun_out/final_latency.txt 2>&1; tail -5 gpurun_out/final_latency.txt
