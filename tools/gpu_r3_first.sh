#!/bin/bash
# Round 3, first GPU call: launch-shape A/B of k_lq<true> (one wave per node vs two) + the GPU tests touched by the ADVICE fixes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
STEPS=10 bash tools/gpu_variants.sh > gpurun_out/r3_first_variants.log 2>&1
timeout 600 python -m pytest tests/test_event_nodes.py tests/test_adaptor.py tests/test_policy.py -m gpu -x -q > gpurun_out/r3_first_tests.log 2>&1
tail -5 gpurun_out/r3_first_tests.log
cat gpurun_out/variants.log
