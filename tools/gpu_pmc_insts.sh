#!/bin/bash
# dynamic instruction mix per kernel (SQ_INSTS_* counters, own pass, no trace domains besides kernel-trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/pmc_insts"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 \
  --kernel-trace --output-format csv -d "$OUT/pmc_insts" -o pmc -- python "$OUT/../bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_insts.log" 2>&1
echo "rc=$?"
cd "$OUT/.."
python - <<'PY'
import csv, glob, re, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_insts/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<\w+>)?)", row["Kernel_Name"])
        if m: per[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in per.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    w = d.get("SQ_WAVES", 1) or 1
    print(k, {c: round(val / w, 1) for c, val in d.items() if c != "SQ_WAVES"}, "waves", int(w), "(instructions per wave)")
PY
