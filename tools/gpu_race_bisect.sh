#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; : > gpurun_out/race.log
for lib in wb_humanoid_mpc_amd/libhsqp_hip.so wb_humanoid_mpc_amd/variants/libhsqp_bis_*.so; do
  HSQP_LIB=$PWD/$lib timeout 200 python tools/race_probe.py ${N:-100} ${GAIT:-walk} ${B:-256} 2>&1 | tail -1 >> gpurun_out/race.log
done
cat gpurun_out/race.log
