#!/bin/bash
# copies the evidence tools/gpu_profiles_r03.sh (and the bench / phase runs named below) left under gpurun_out/ into profiles/r03_*
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/kernel_stats.csv $P/r03_kernel_stats.csv; cp $G/pmc_summary.json $P/r03_pmc_summary.json; cp $G/pmc_sq_summary.txt $P/r03_pmc_sq_summary.txt
cp $G/pmc_sq_summary.json $P/r03_pmc_sq_summary.json; cp $G/pmc_insts_summary.txt $P/r03_pmc_insts_summary.txt
cp $G/pmc_sq_counter_collection.csv $P/r03_pmc_sq_counter_collection.csv; cp $G/pmc_insts_counter_collection.csv $P/r03_pmc_insts_counter_collection.csv
cp $G/pmc_FETCH_SIZE/pmc_counter_collection.csv $P/r03_pmc_FETCH_SIZE_counter_collection.csv; cp $G/pmc_WRITE_SIZE/pmc_counter_collection.csv $P/r03_pmc_WRITE_SIZE_counter_collection.csv
for f in bench_wb bench_cfg3 bench_cfg3_serial bench_cfg5 bench_cent_cfg1 bench_cent_cfg2 bench_strong32 bench_strong64; do [ -f $G/$f.log ] && cp $G/$f.log $P/r03_$f.json; done
[ -f $G/phase.log ] && cp $G/phase.log $P/r03_phase_profile.txt
[ -f $G/parity_report.json ] && cp $G/parity_report.json $P/r03_parity_report.json && cp $G/parity_report.log $P/r03_parity_report.txt
[ -f $G/resource_usage.txt ] && cp $G/resource_usage.txt $P/r03_resource_usage.txt
ls $P | grep r03
