#!/bin/bash
# (GPU) A/B of the whole-body LQ approximation: limb-lane form (k_lq_limb + k_lq_rows + k_lq_chain, hsqp_lql.h) against the phase form (k_lq<true>,
# HSQP_LQ_PHASE_FORM=1) and against every library under wb_humanoid_mpc_amd/variants/: parity subset first, then bench lines, then the
# rocprofv3 kernel stats of the product library (the split between the two limb-form kernels).
# Usage: gpurun -- 'TESTS="tests/test_gpu_parity.py -k lq_blocks" bash tools/gpu_lq_ab.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
line() { python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt", d.get("kkt_over_max_1_g_inf"))
    elif "rror" in line: print(line[:300])
'; }
{
libs=("" $(ls wb_humanoid_mpc_amd/variants/libhsqp_*.so 2>/dev/null | sed 's/.*libhsqp_//; s/\.so//'))
for v in "${libs[@]}"; do
  lib=$PWD/wb_humanoid_mpc_amd/libhsqp_hip.so; [ -n "$v" ] && lib=$PWD/wb_humanoid_mpc_amd/variants/libhsqp_$v.so
  echo "== ${v:-product}"
  [ -n "$TESTS" ] && HSQP_LIB=$lib timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -5
  for rep in 1 2; do HSQP_LIB=$lib timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | line; done
done
echo "== product, phase form (HSQP_LQ_PHASE_FORM=1)"
for rep in 1 2; do HSQP_LQ_PHASE_FORM=1 timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | line; done
for shape in "32 100" "64 100" "128 100"; do
  set -- $shape
  echo "== strong leg $1 x $2: limb / phase"
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch $1 --nodes $2 2>&1 | line
  HSQP_LQ_PHASE_FORM=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch $1 --nodes $2 2>&1 | line
done
echo "== rocprofv3 kernel stats (product)"
rm -rf "$OUT/prof_lq"
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lq" -o bench -- python $OUT/../bench.py --no-cpu-baseline --steps 5 --warmup 1 > "$OUT/prof_lq.log" 2>&1 )
find "$OUT/prof_lq" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/lq_kernel_stats.csv"
head -14 "$OUT/lq_kernel_stats.csv" | cut -c1-170
} > gpurun_out/lq_ab.log 2>&1
cat gpurun_out/lq_ab.log
