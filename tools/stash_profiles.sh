#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/final_<precision>/ into profiles/ (tracked):
#   bash tools/stash_profiles.sh r04_z
set -e
cd "$(dirname "$0")/.."
TAG=${1:?tag}
for pair in fp32_split:split bf16a:bf16a bf16:bf16 fp32:fp32; do
  P=${pair%%:*}; N=${pair##*:}; O=gpurun_out/final_$P
  [ -f $O/bench.json ] && [ -d $O/stats_onestream ] || continue
  cp $O/bench.json profiles/${TAG}_${N}_bench.json
  cp $(ls -t $(find $O/stats -name "*kernel_stats.csv") | head -1) profiles/${TAG}_${N}_kernel_stats.csv
  cp $(ls -t $(find $O/stats_onestream -name "*kernel_stats.csv") | head -1) profiles/${TAG}_${N}_onestream_kernel_stats.csv
  cp $O/pmc_per_kernel.csv profiles/${TAG}_${N}_pmc_per_kernel.csv
  cp $O/traffic.json profiles/${TAG}_${N}_traffic.json
  if [ -f $O/frame256_traffic.json ]; then
    cp $O/frame256_traffic.json profiles/${TAG}_${N}_frame256_traffic.json
    cp $O/frame256_pmc_per_kernel.csv profiles/${TAG}_${N}_frame256_pmc_per_kernel.csv
  fi
  echo "stashed $P -> profiles/${TAG}_${N}_*"
done
