#!/bin/bash
# GPU parity tests against every tuning build under wb_humanoid_mpc_amd/variants/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in wb_humanoid_mpc_amd/variants/libhsqp_*.so; do
  echo "== $lib"; HSQP_LIB=$PWD/$lib timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
done
