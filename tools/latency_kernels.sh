#!/bin/bash
# developer probe: per-kernel durations of the replayed single-system step (64 atoms unless N_ATOMS is set), through gpurun
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/lat
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lat -- python $R/tools/latency_profile.py > /dev/null 2>&1
python - "$R" <<'PY'
import csv, glob, os, sys
p = sorted(glob.glob(os.path.join(sys.argv[1], "gpurun_out/lat/**/*kernel_stats.csv"), recursive=True))[-1]
tot = 0.0
n = 0.0
for r in csv.DictReader(open(p)):
    calls = int(r["Calls"])
    if calls < 150:
        continue
    per = calls / 201.0
    tot += float(r["TotalDurationNs"]) / 201.0
    n += per
    print("%7.2f us x%4.1f  %s" % (float(r["AverageNs"]) / 1e3, per, r["Name"][:110]))
print("kernels per step: %.1f, sum of their durations: %.1f us" % (n, tot / 1e3))
PY
