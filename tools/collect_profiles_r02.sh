#!/bin/bash
# copies the evidence tools/gpu_profiles_r02.sh, gpu_bench.sh, gpu_r2_final.sh and gpu_scan_profile.sh left under gpurun_out/ into profiles/r02_*
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/kernel_stats.csv $P/r02_kernel_stats.csv; cp $G/pmc_summary.json $P/r02_pmc_summary.json; cp $G/pmc_sq_summary.txt $P/r02_pmc_sq_summary.txt
cp $G/pmc_sq_counter_collection.csv $P/r02_pmc_sq_counter_collection.csv
cp $G/pmc_FETCH_SIZE/pmc_counter_collection.csv $P/r02_pmc_FETCH_SIZE_counter_collection.csv; cp $G/pmc_WRITE_SIZE/pmc_counter_collection.csv $P/r02_pmc_WRITE_SIZE_counter_collection.csv
cp $G/bench_wb.log $P/r02_bench.json; cp $G/bench_cfg3.log $P/r02_bench_cfg3.json; cp $G/bench_cent_cfg2.log $P/r02_bench_cent_cfg2.json; cp $G/bench_strong1.log $P/r02_bench_datapath_one_gpu.json
cp $G/phase.log $P/r02_phase_profile.txt; cp $G/phase_b1.log $P/r02_phase_profile_b1.txt; cp $G/parity_report.json $P/r02_parity_report.json; cp $G/parity_report.log $P/r02_parity_report.txt
cp $G/bench_cfg5.log $P/r02_bench_cfg5.json; cp $G/scan_wb_kernel_stats.csv $P/r02_scan_wb_kernel_stats.csv; cp $G/scan_cent_kernel_stats.csv $P/r02_scan_cent_kernel_stats.csv
cp $G/bench_cent_cfg1.log $P/r02_bench_cent_cfg1.json; cp $G/bench_cfg3_serial.log $P/r02_bench_cfg3_serial.json
