#!/bin/bash
# rocprofv3 kernel stats of the two-level sweep at 32 instances (one GPU of an 8-GPU run of config 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
rm -rf "$OUT/prof_seg"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_seg" -o seg -- python "$OUT/../bench.py" --no-cpu-baseline --batch ${SEGB:-32} --riccati ${SEGR:-segmented} --steps 5 --warmup 1 > "$OUT/prof_seg.log" 2>&1
find "$OUT/prof_seg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/seg_kernel_stats.csv"
python - <<'PY'
import csv, re
for row in csv.DictReader(open("/root/repo/gpurun_out/seg_kernel_stats.csv")):
    m = re.search(r"(k_\w+(?:<\w+>)?)", row["Name"])
    if m: print(f"{m.group(1):28s} calls {row['Calls']:>4s} avg us {float(row['AverageNs'])/1e3:9.1f}  total % {row['Percentage']}")
PY
