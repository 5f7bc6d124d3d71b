#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC passes -> gpurun_out/final/
set -x
# usage: collect_profiles.sh [precision]   (fp32_split | fp32 | bf16 | bf16a)
PREC=${1:-fp32_split}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$PREC; rm -rf $O; mkdir -p $O
cd $R && python bench.py --precision $PREC --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --precision $PREC --steps 20 --warmup 3 --no-cpu-baseline --no-profile > $O/stats.log 2>&1
# the same on ONE stream (the batch un-split, as bench.py's per-kernel profiled pass runs it): launch durations that do not
# overlap with the other sub-batch's kernels -- these are the averages that bench.py's roofline objects must agree with
export ACX_SPLIT_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_onestream -- python3 $R/bench.py --precision $PREC --steps 20 --warmup 3 --no-cpu-baseline --no-profile > $O/stats_onestream.log 2>&1
unset ACX_SPLIT_STREAMS
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/tools/prof_step.py --precision $PREC > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/prof_step.py --precision $PREC > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/prof_step.py --precision $PREC > $O/pmc_write.log 2>&1
cd $R && python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_per_kernel.csv $O/traffic.json $PREC > $O/pmc_table.log 2>&1
tail -3 $O/pmc_table.log; cat $O/bench.json | cut -c1-400
if [ "$PREC" = fp32_split ] && [ "${FRAME256:-1}" = 1 ]; then
  # BASELINE configs[3] (forward_frame_embeddings, bs = 256): the counter traffic of ITS launches (bench.py: frame_bs256.*.traffic)
  cd /tmp
  F="--precision fp32_split --batch 256 --mode frame"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/f256_sq -- python3 $R/tools/prof_step.py $F > $O/f256_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f256_fetch -- python3 $R/tools/prof_step.py $F > $O/f256_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/f256_write -- python3 $R/tools/prof_step.py $F > $O/f256_write.log 2>&1
  cd $R && python tools/pmc_table.py $O/f256_sq $O/f256_fetch $O/f256_write $O/frame256_pmc_per_kernel.csv $O/frame256_traffic.json fp32_split "one forward_frame_embeddings, B=256, 10 s @ 32 kHz" > $O/f256_table.log 2>&1
  tail -3 $O/f256_table.log
fi
