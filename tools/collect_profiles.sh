#!/bin/bash
# copies the evidence tools/gpu_round_final.sh (bench lines, parity report, phase profile, tools/gpu_profiles.sh) left under gpurun_out/ into
# profiles/<round>_*   (usage: tools/collect_profiles.sh r05)
cd "$(dirname "$0")/.."
R=${1:-r05}; G=gpurun_out; P=profiles
cp $G/kernel_stats.csv $P/${R}_kernel_stats.csv; cp $G/pmc_summary.json $P/${R}_pmc_summary.json; cp $G/pmc_sq_summary.txt $P/${R}_pmc_sq_summary.txt
cp $G/pmc_sq_summary.json $P/${R}_pmc_sq_summary.json; cp $G/pmc_insts_summary.txt $P/${R}_pmc_insts_summary.txt
cp $G/pmc_sq_counter_collection.csv $P/${R}_pmc_sq_counter_collection.csv; cp $G/pmc_insts_counter_collection.csv $P/${R}_pmc_insts_counter_collection.csv
cp $G/pmc_FETCH_SIZE/pmc_counter_collection.csv $P/${R}_pmc_FETCH_SIZE_counter_collection.csv; cp $G/pmc_WRITE_SIZE/pmc_counter_collection.csv $P/${R}_pmc_WRITE_SIZE_counter_collection.csv
[ -s $G/strong_prediction.json ] && cp $G/strong_prediction.json $P/${R}_strong_prediction.json
for f in bench_wb bench_cfg3 bench_cfg3_serial bench_cfg5 bench_cent_cfg1 bench_cent_cfg2 bench_strong32 bench_strong64 bench_strong128 bench_strong128_n200; do [ -f $G/$f.log ] && cp $G/$f.log $P/${R}_$f.json; done
[ -f $G/phase.log ] && cp $G/phase.log $P/${R}_phase_profile.txt
[ -f $G/parity_report.json ] && cp $G/parity_report.json $P/${R}_parity_report.json && cp $G/parity_report.log $P/${R}_parity_report.txt
[ -f $G/resource_usage.txt ] && cp $G/resource_usage.txt $P/${R}_resource_usage.txt
ls $P | grep $R
