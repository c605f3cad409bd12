#!/bin/bash
# round-2 evidence: rocprofv3 kernel stats of the bench command, HBM traffic counters (separate passes, one counter each), SQ counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
rm -rf "$OUT/prof" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_sq"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$OUT/../bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/prof.log" 2>&1
echo "rocprof stats exit: $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- python "$OUT/../bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$C.log" 2>&1
  echo "rocprof $C exit: $?"
done
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- python "$OUT/../bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
echo "rocprof sq exit: $?"
cd "$OUT/.."
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/pmc_sq" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/pmc_sq_counter_collection.csv"
python - <<'PY' > gpurun_out/pmc_sq_summary.txt
import csv, glob, re, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<\w+>)?)", row["Kernel_Name"])
        if m: per[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in per.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "(M per launch)")
PY
cat gpurun_out/pmc_sq_summary.txt; head -12 "$OUT/kernel_stats.csv" | cut -c1-160; cat "$OUT/pmc_summary.json" | head -40
