"""Developer helper: the heaviest kernels of a rocprofv3 --kernel-trace --stats output directory.  usage: python tools/kstats_top.py <dir> [n]"""
import csv, glob, sys
p = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(p)))[:n]:
    print("%8.2f ms %6d x %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
