#!/bin/bash
# final-build evidence: bit-identical repeats (all four arithmetics, three shapes), step-time spread, packed-FP32 victim detector beside the shipped kernels
export GPU_MAX_HW_QUEUES=2
python tools/soak.py > gpurun_out/final_soak.txt 2>&1; cat gpurun_out/final_soak.txt | tail -14
python tools/step_jitter.py fp32_split 400 > gpurun_out/final_jitter.txt 2>&1; python tools/step_jitter.py bf16a 400 >> gpurun_out/final_jitter.txt 2>&1; grep -E "median|p99|over" gpurun_out/final_jitter.txt | head -8
