"""Issue-side counters of the dominant kernels ("not bound by bytes needs its own measured ceiling").
Run on the GPU box:  python tools/pmc_issue.py [out.json]
Several rocprofv3 --pmc passes (own runs, --kernel-trace only - never combined with other trace domains) over the headline bench
leg; counters are filtered by `rocprofv3 -L` first, so a name this ROCm build lacks costs nothing.  Per kernel (largest grid): mean
per launch of every counter + derived fractions:
  wait_any / wave_cycles            waves parked on s_waitcnt / barriers
  wait_inst_any / wave_cycles       issue stalls (pipe busy, dependency)
  active_vmem, active_lds / wave_cycles
  vmem instructions per wave, LDS bank-conflict cycles / LDS active cycles, TA busy, TCP pending-stall cycles
and the REQUEST ceiling of the sweeps: measured L2 requests per launch / the 128-byte-request rate tools/microbench/gather_bw.hip
sustains (6.8 TB/s of 128-byte pieces = 53 G requests/s) - the second roofline bench.py prints next to the HBM one."""
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_pmc_issue.json")
WISH = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VMEM", "SQ_LDS_ADDR_CONFLICT"],
    ["TA_TA_BUSY_sum", "TA_BUSY_avr", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_GATE_EN1_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
]
KRE = re.compile(os.environ.get("KREGEX", r"k_message_rows8|k_message_adjoint_rows8|k_tlin9|k_edge_interp|k_message_bwd_node"))
env = dict(os.environ, TMPDIR="/tmp")
try:
    avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120, env=env).stdout
except Exception as e:  # noqa: BLE001
    avail = ""
names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", avail))
res = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
dropped, passes = [], []
for i, group in enumerate(WISH):
    use = [c for c in group if not names or c in names]
    dropped += [c for c in group if c not in use]
    if not use:
        continue
    d = f"/tmp/pmc_issue_{i}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc", *use, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-md", "--no-aux"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env)
    passes.append({"counters": use, "rc": r.returncode})
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("tn::", "").strip()
            if KRE.search(name):
                res[name][row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
out = {"_what": __doc__.split("\n")[0], "command": "bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md --no-aux (C2: 256 x 64 atoms)",
       "git_commit": os.environ.get("GIT_COMMIT", "unknown"), "passes": passes, "counters_not_in_this_rocm": dropped, "kernels": {}}
for name, cnt in sorted(res.items()):
    k = {}
    for c, lst in sorted(cnt.items()):
        g = max(x for x, _ in lst)
        v = [y for x, y in lst if x == g]
        k[c] = sum(v) / len(v)
        k["_launches_averaged"] = len(v)
    wc = k.get("SQ_WAVE_CYCLES")
    if wc:
        for key, c in (("frac_wait_any", "SQ_WAIT_ANY"), ("frac_wait_inst_any", "SQ_WAIT_INST_ANY"), ("frac_active_inst_any", "SQ_ACTIVE_INST_ANY"),
                       ("frac_active_vmem", "SQ_ACTIVE_INST_VMEM"), ("frac_active_lds", "SQ_ACTIVE_INST_LDS"), ("frac_active_valu", "SQ_ACTIVE_INST_VALU")):
            if c in k:
                k[key] = k[c] / wc
    if k.get("SQ_LDS_IDX_ACTIVE"):
        k["lds_bank_conflict_frac"] = k.get("SQ_LDS_BANK_CONFLICT", 0.0) / k["SQ_LDS_IDX_ACTIVE"]
    if k.get("TCC_HIT_sum") is not None and k.get("TCC_MISS_sum") is not None and k["TCC_HIT_sum"] + k["TCC_MISS_sum"] > 0:
        k["l2_hit_rate"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
    out["kernels"][name] = k
os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps({n: {c: round(v, 4) for c, v in k.items() if c.startswith("frac") or c in ("l2_hit_rate", "lds_bank_conflict_frac")} for n, k in out["kernels"].items()}, indent=1))
