#!/bin/bash
# developer A/B: the headline step and the per-launch time of every kernel group, once per environment setting given as arguments
# usage (GPU box): tools/ab_bench.sh "X=1" "TMDNET_FOO=1" ...
R=${GRAFT_REPO_ROOT:-$PWD}
for e in "$@"; do
  env $e python $R/bench.py --no-cpu-baseline --no-md --no-aux --breakdown $R/gpurun_out/ab_bd.json > $R/gpurun_out/ab_b.json 2>/dev/null
  python - "$e" "$R" <<'PY'
import json, sys
e, R = sys.argv[1:3]
d = json.loads(open(f"{R}/gpurun_out/ab_b.json").read().strip().splitlines()[-1]); b = json.load(open(f"{R}/gpurun_out/ab_bd.json"))
print(e, "ms_per_step", round(d["ms_per_step"], 4), {k: round(v["ms"], 3) for k, v in b["one_step_profiled_ms"].items()})
print("   ", {k.split(": ")[1][:34]: round(v["ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in b["kernels"].items() if k.startswith("gemm_node: tlin9") or k.startswith("message")})
PY
done
