cd "${GRAFT_REPO_ROOT:-/root/repo}"
for B in 2 3 4 6 8 12 16; do
  for r in serial parallel; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch $B --riccati $r --no-strong --sustained 0 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], sys.argv[2], "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt", d.get("kkt_over_max_1_g_inf"), "fallbacks", d.get("scan_fallbacks"))
' $B $r
  done
done
