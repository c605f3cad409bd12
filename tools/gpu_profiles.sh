#!/bin/bash
# evidence of a round (GPU box, via gpurun): rocprofv3 kernel stats of the bench command, HBM traffic counters (separate --pmc passes, one
# counter each, kernel trace only), SQ counters, instruction mix; summaries under gpurun_out/ — tools/collect_profiles.sh copies them
# to profiles/<round>_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
rm -rf "$OUT/prof" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_sq" "$OUT/pmc_insts"
cd /tmp && export TMPDIR=/tmp
BENCH="python $OUT/../bench.py --no-cpu-baseline --sustained 0"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- $BENCH --steps 5 --warmup 1 > "$OUT/prof.log" 2>&1
echo "rocprof stats exit: $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_$C.log" 2>&1
  echo "rocprof $C exit: $?"
done
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
echo "rocprof sq exit: $?"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES \
  --kernel-trace --output-format csv -d "$OUT/pmc_insts" -o pmc -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_insts.log" 2>&1
echo "rocprof insts exit: $?"
cd "$OUT/.."
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/pmc_sq" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/pmc_sq_counter_collection.csv"
find "$OUT/pmc_insts" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/pmc_insts_counter_collection.csv"
python tools/pmc_sq_summary.py "$OUT"
cat "$OUT/pmc_sq_summary.txt" "$OUT/pmc_insts_summary.txt"; head -12 "$OUT/kernel_stats.csv" | cut -c1-160; head -60 "$OUT/pmc_summary.json"
