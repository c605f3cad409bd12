#!/bin/bash
# phase ticks with one workgroup per CU (256 nodes) vs the full bench shape: separates latency from contention
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/phase_profile.py 2 128 > gpurun_out/phase_small.log 2>&1
timeout 300 python tools/phase_profile.py 256 100 > gpurun_out/phase.log 2>&1
grep -v "  [0-1]\.[0-9]%" gpurun_out/phase_small.log | head -70
