"""Turn rocprofv3 --pmc counter CSVs into per-launch HBM traffic per kernel class (profiles/pmc_traffic.json).

Usage (on the GPU box; FETCH_SIZE and WRITE_SIZE need separate passes, MI355X_MICROARCH.md "PMC slots"):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_rd -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_wr -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md
  python tools/pmc_traffic.py gpurun_out/pmc_rd gpurun_out/pmc_wr profiles/pmc_traffic.json

Units/corrections: FETCH_SIZE and WRITE_SIZE are reported in KiB... the guide's gfx950 note says FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so reads are doubled; WRITE_SIZE is taken as reported (uncalibrated).
Only dispatches of the bench workload's grid (the largest grid of each kernel) are averaged, so the small
single-system MD leg of bench.py does not dilute the figure.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import re

# kernel (rocprofv3 short name, template instance kept) -> profiled class of the library (tn_model.h ProfCat).  Exact
# patterns: `k_message_bwd_node` is an elementwise kernel, not a sweep.
CLASS_OF = [(r"k_gemm_dual", "gemm_edge"), (r"k_edge_mlp", "gemm_edge"), (r"k_gemm_(nt|skinny|sb1)", "gemm_node"), (r"k_tlin9", "gemm_node"),
            (r"k_message(_tile|_rows8|_adjoint|_adjoint_gd|_split)?(<.*>)?$", "message"),
            (r"k_(edge_interp|pair_cutoff_hist|bucket_scan|bucket_scatter)", "edge_table"),
            (r"k_(pair_gd|embed_pair_gd|geom_gd|embed_gm|embed_pair_rb)", "pair_bwd"),
            (r"k_embed_(scatter|moments|combine)", "embed_scatter")]


def read_counter(dirname, counter):
    rows = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                rows.append((r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    return rows


def per_kernel(rows):
    by = defaultdict(list)
    for name, grid, val in rows:
        short = name.split("(")[0].replace("void ", "").replace("tn::", "").strip()  # "k_gemm_dual_sb2<2>"
        by[short].append((grid, val, name))
    out = {}
    for k, lst in by.items():
        gmax = max(g for g, _, _ in lst)
        vals = [v for g, v, _ in lst if g == gmax]
        out[k] = {"launches": len(vals), "mean": sum(vals) / len(vals), "grid": gmax}
    return out


def main():
    rd_dir, wr_dir, out_path = sys.argv[1:4]
    rd = per_kernel(read_counter(rd_dir, "FETCH_SIZE"))
    wr = per_kernel(read_counter(wr_dir, "WRITE_SIZE"))
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        r = rd.get(k, {}).get("mean", 0.0) * 1024.0 * 2.0  # KiB -> B, gfx950 half-count correction
        w = wr.get(k, {}).get("mean", 0.0) * 1024.0
        kernels[k] = {"read_bytes_per_launch": r, "write_bytes_per_launch": w, "launches_seen": rd.get(k, wr.get(k))["launches"],
                      "grid": rd.get(k, wr.get(k))["grid"]}
    classes = defaultdict(lambda: [0.0, 0])
    for k, v in kernels.items():
        for pat, cls in CLASS_OF:
            if re.match(pat, k):
                n = v["launches_seen"]
                classes[cls][0] += (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * n
                classes[cls][1] += n
                break
    res = {cls: tot / max(n, 1) for cls, (tot, n) in classes.items()}  # mean HBM bytes per launch of the class
    res["_per_kernel"] = kernels
    res["_per_kernel_total"] = {k: v["read_bytes_per_launch"] + v["write_bytes_per_launch"] for k, v in kernels.items()}
    res["_meta"] = {"git_commit": os.environ.get("GIT_COMMIT"), "recipe": "tools/profile_round.sh (two rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE)",
                    "bench_command": sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md --no-aux"}
    res["_note"] = "bytes per launch; reads = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), writes = WRITE_SIZE KiB x 1024"
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    for cls, (tot, n) in classes.items():
        print(f"{cls:14s} {tot / max(n, 1) / 1e6:10.2f} MB/launch over {n} launches")


if __name__ == "__main__":
    main()
