#!/bin/bash
# Round 6: phase profiles of the timing-experiment builds of k_riccati_fact (wb_humanoid_mpc_amd/variants/libhsqp_*.so, -DHSQP_EXP=<mask>: WRONG results,
# timings only).  Usage: gpurun -- 'bash tools/gpu_r06_exp.sh [B]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r06_exp.log
: > $L
for lib in wb_humanoid_mpc_amd/variants/libhsqp_*.so; do
  v=$(basename $lib .so | sed 's/libhsqp_//')
  echo "== $v" >> $L
  HSQP_PROF_LIB=$PWD/$lib timeout 300 python tools/phase_profile.py ${1:-256} 100 2>&1 | sed -n '/kernel ms/p; /k_riccati/,/Ph3 arrival/p' >> $L
done
cat $L
