#!/bin/bash
# usage: tools/publish_profiles.sh [round = r06]  -- copies what tools/collect_round.sh merged into gpurun_out/<round>/ and gpurun_out/raw_<round>/
# to profiles/ (tracked), rebuilds the roofline json from the raw CSVs and prints the figures the docs quote.
set -e
cd "$(dirname "$0")/.."
R=${1:-r06}
for f in gpurun_out/$R/*.txt; do grep -v "amdgpu.ids" "$f" > profiles/${R}_$(basename "$f"); done
cp gpurun_out/$R/bench_line.json profiles/${R}_bench_line.json
rm -rf profiles/raw_$R; mkdir -p profiles/raw_$R
cp gpurun_out/raw_$R/kt_kernel_stats.csv gpurun_out/raw_$R/pmc_*_counter_collection.csv gpurun_out/raw_$R/workload.json profiles/raw_$R/
cp profiles/raw_$R/kt_kernel_stats.csv profiles/${R}_kernel_stats_bench_B131072.csv
python tools/roofline.py rebuild --raw profiles/raw_$R --out profiles/${R}_roofline_bench.json > /dev/null
python tools/roofline.py show profiles/${R}_roofline_bench.json profiles/${R}_bench_line.json | tail -8
