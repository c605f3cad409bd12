#!/bin/bash
# Per-phase tick profile only (via gpurun): tools/phase_profile.py with the -DHSQP_PHASE_PROFILE build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/phase_profile.py ${PB:-256} ${PN:-100} > gpurun_out/phase.log 2>&1
sed -n '/k_riccati/,/k_lq<false>/p' gpurun_out/phase.log
