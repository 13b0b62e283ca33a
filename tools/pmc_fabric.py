"""Memory-side counters of the large-system neighbour sweeps ("is FETCH_SIZE counting memory-side-cache hits?
if HBM bytes are already ~1x the kernel is fabric-bound - say so with the counter").  Run on the GPU box:
    python tools/pmc_fabric.py [out.json]
Separate rocprofv3 --pmc passes (--kernel-trace only) over tools/water_profile.py; counter names are filtered by `rocprofv3 -L`,
whose lines matching EA / DRAM / MALL / HBM are kept in the output so that what this ROCm build can and cannot see is on record.
Per kernel (largest grid, mean per launch): L2 requests, hits, misses, requests the L2 sent to the fabric (TCC_EA0_RDREQ: 64-byte
units unless _32B), and - where the build has them - the share of those that went to DRAM."""
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_fabric.json")
TARGET = os.environ.get("TARGET", os.path.join(ROOT, "tools", "water_profile.py"))
WISH = [
    ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_READ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    ["TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_RD_UNCACHED_32B_sum", "TCC_EA0_WR_UNCACHED_32B_sum"],
    ["TCC_EA0_RDREQ_IO_CREDIT_STALL_sum", "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_TAG_STALL_sum"],
    ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_LEVEL_avr", "TCC_BUSY_sum", "TCC_CYCLE_sum"],
    ["FETCH_SIZE"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TA_TA_BUSY_sum"],
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"],
    ["TA_BUSY_avr", "TA_BUSY_max", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TA_TCP_STATE_READ_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]
if os.environ.get("PASSES"):  # e.g. PASSES=0,4,6,7,8 : a subset of the passes above
    WISH = [WISH[int(i)] for i in os.environ["PASSES"].split(",")]
KRE = re.compile(os.environ.get("KREGEX", r"k_message_adjoint_gd|k_message<|k_edge_interp|k_tlin9|k_gemm_sb1"))
env = dict(os.environ, TMPDIR="/tmp")
try:
    avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=180, env=env).stdout
except Exception:  # noqa: BLE001
    avail = ""
names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", avail))
listing = sorted({n for n in names if re.search(r"EA0?_|DRAM|MALL|HBM|FETCH|WRITE_SIZE|GMI", n)})
res = defaultdict(lambda: defaultdict(list))
durs = defaultdict(list)
dropped, passes = [], []
for i, group in enumerate(WISH):
    use = [c for c in group if not names or c in names]
    dropped += [c for c in group if c not in use]
    if not use:
        continue
    d = f"/tmp/pmc_fabric_{i}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc", *use, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, TARGET]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env)
    passes.append({"counters": use, "rc": r.returncode, "tail": (r.stdout + r.stderr)[-200:] if r.returncode else ""})
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("tn::", "").strip()
            if KRE.search(name):
                res[name][row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("tn::", "").strip()
            if KRE.search(name):
                durs[name].append((int(row.get("Grid_Size", 0) or 0), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
out = {"_what": __doc__.split("\n")[0], "target": os.path.basename(TARGET), "git_commit": os.environ.get("GIT_COMMIT", "unknown"),
       "passes": passes, "counters_not_in_this_rocm": dropped, "memory_side_counters_this_rocm_lists": listing, "kernels": {}}
for name, cnt in sorted(res.items()):
    k = {}
    for c, lst in sorted(cnt.items()):
        g = max(x for x, _ in lst)
        v = [y for x, y in lst if x == g]
        k[c] = sum(v) / len(v)
    if durs.get(name):
        g = max(x for x, _ in durs[name])
        v = [y for x, y in durs[name] if x == g]
        k["avg_us_under_counters"] = sum(v) / len(v)
    if k.get("TCC_HIT_sum") is not None and k.get("TCC_MISS_sum") is not None and k["TCC_HIT_sum"] + k["TCC_MISS_sum"] > 0:
        k["l2_hit_rate"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
    if k.get("TCC_EA0_RDREQ_sum") is not None:
        r32 = k.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        k["fabric_read_bytes"] = (k["TCC_EA0_RDREQ_sum"] - r32) * 64 + r32 * 32
        if k.get("TCC_EA0_RDREQ_DRAM_sum") is not None and k["TCC_EA0_RDREQ_sum"] > 0:
            k["dram_share_of_fabric_reads"] = k["TCC_EA0_RDREQ_DRAM_sum"] / k["TCC_EA0_RDREQ_sum"]
    out["kernels"][name] = k
os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps({"listed": listing, "dropped": dropped,
                  "kernels": {n: {c: (round(v, 4) if isinstance(v, float) else v) for c, v in k.items()} for n, k in out["kernels"].items()}}, indent=1)[:6000])
