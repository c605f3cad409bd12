cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3 4; do for v in "" diet0 diet1; do
 lib=$PWD/wb_humanoid_mpc_amd/libhsqp_hip.so; [ -n "$v" ] && lib=$PWD/wb_humanoid_mpc_amd/variants/libhsqp_$v.so
 HSQP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 3), {k: round(v, 4) for k, v in d["kernel_ms"].items()})
' "${v:-product}"
done; done
