#!/bin/bash
# SQ counters of the bench kernels (MFMA busy, wait buckets, LDS conflicts): one rocprofv3 --pmc pass, kernel trace only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- python "$OUT/../bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
echo "rocprof exit: $?" >> "$OUT/pmc_sq.log"
cd "$OUT/.."
python - <<'PY'
import csv, glob, re, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<\w+>)?)", row["Kernel_Name"])
        if m: per[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in per.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "(M per launch)")
PY
tail -3 "$OUT/pmc_sq.log"
