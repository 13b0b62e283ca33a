#!/bin/bash
# Same-box A/B of developer switches: runs the headline bench leg once per environment setting and prints step time and the
# per-class breakdown.  usage (through gpurun):  tools/ab_env.sh "" "TMDNET_X=1" "TMDNET_Y=1 TMDNET_Z=2" ...
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/ab
k=0
for envs in "$@"; do
  for rep in 1 2; do
    out=$R/gpurun_out/ab/run${k}_${rep}
    env $envs python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-md --no-aux --breakdown $out.bd.json > $out.json 2> $out.err
    python - "$out" "$envs" <<'PY'
import json, sys
out, envs = sys.argv[1], sys.argv[2]
try:
    line = json.loads(open(out + ".json").read().strip().splitlines()[-1])
    bd = json.load(open(out + ".bd.json"))
    cls = {k: round(v["ms"], 3) for k, v in bd["one_step_profiled_ms"].items()}
    print(f"[{envs or 'default'}] ms_per_step={line['ms_per_step']:.4f} classes={cls}")
    for k, v in bd["kernels"].items():
        if v["ms"] > 0.03 and ("message" in k or "tensor_linear" in k or "tlin9" in k):
            print(f"    {v['ms']*1000/v['launches']:8.1f} us x{v['launches']}  {k[:110]}")
except Exception as e:
    print(f"[{envs}] failed: {e}")
    print(open(out + ".err").read()[-2000:])
PY
  done
  k=$((k+1))
done
