#!/bin/bash
# Counter probe for A/B work (GPU box, through gpurun): one rocprofv3 --pmc pass per (environment, counter set) over the headline
# bench leg, then the per-kernel mean of every counter for kernels matching $KREGEX (largest grid only).
# usage: KREGEX='k_message' tools/pmc_probe.sh "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" -- "" "TMDNET_X=1"
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
sets=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done
shift
cd /tmp
k=0
for envs in "$@"; do
  j=0
  for cs in "${sets[@]}"; do
    d=$R/gpurun_out/probe/e${k}_s${j}
    rm -rf $d; mkdir -p $d
    env $envs rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $d -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md --no-aux > $d/log.txt 2>&1
    python - "$d" "$envs" "$cs" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
d, envs, cs = sys.argv[1:4]
rx = re.compile(os.environ.get("KREGEX", "."))
by = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tn::", "").strip()
        if rx.search(name):
            by[name][r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
for name, cnt in sorted(by.items()):
    out = []
    for c, lst in sorted(cnt.items()):
        g = max(x for x, _ in lst)
        v = [y for x, y in lst if x == g]
        out.append(f"{c}={sum(v)/len(v):.4g}")
    print(f"[{envs or 'default'}] {name[:60]}: " + " ".join(out))
PY
    find $d -name '*.csv' -size +2M -delete
    j=$((j+1))
  done
  k=$((k+1))
done
