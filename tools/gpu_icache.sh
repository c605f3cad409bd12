#!/bin/bash
# instruction-cache counters of the bench command (GPU box): are the big straight-line kernels fetch-bound?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
rm -rf "$OUT/pmc_icache"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d "$OUT/pmc_icache" -o pmc -- python "$OUT/../bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/pmc_icache.log" 2>&1
echo "rc=$?"
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("/root/repo/gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True)
if not f: print(open("/root/repo/gpurun_out/pmc_icache.log").read()[-1500:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    m = re.search(r"(k_\w+(?:<\w+>)?)", r["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[m.group(1)] += 1
for k, v in acc.items():
    d = max(n[k], 1)
    print(k, {c: round(x / d / 1e6, 3) for c, x in v.items()}, "(M per launch)")
PY
