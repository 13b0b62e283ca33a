"""bench.py -- molecules/s of the TensorNet energy+force path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

With N > 1 and no torch.distributed environment the script launches its own N ranks (one per GPU) through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; launched by the driver under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.

One "step" = one full pass of the hot path over one batch resident in HBM: neighbour graph build, radial functions,
embedding, L interaction layers, readout, per-molecule energies AND the hand-written reverse pass for the forces
(BASELINE configs[1]: TensorNet 128-hidden 2-layer, synthetic 64-atom x 256-molecule batch, fp32).
  weak   (default): every rank evaluates its own 256 molecules, no data-path collective, one RCCL all-reduce of the
                    zero-padded per-molecule energy vector per step;
  strong (BASELINE configs[2]): the SAME 256-molecule batch is sharded over the ranks by contiguous molecule ranges
                    (torchmdnet_amd.parallel.ShardedEvaluator: forces stay local, energies all-reduced over RCCL).
With N > 1 the other mode is timed as well and reported in a sub-object.

Prints ONE JSON line (rank 0).  `roofline` = the dominant KERNEL (one kernel on one shape) of the dominant kernel class,
measured live with HIP events recorded by the library on the launch stream around every launch during the timed
region, with the class average beside it; `roofline_scatter` = the CSR message sweeps against the HBM peak;
`cpu_baseline` = the oracle (a restatement of the reference's PyTorch CPU path) timed on this host; `et_c4` and
`water10k` = BASELINE configs[3] / configs[4] with their own dominant-kernel rooflines; `tensornet2` = the AceFF-2.0
architecture at the same batch; `md_single_system` = ns/day of
a HIP-graph-replayed 64-atom system; `one_system_8_slabs` = one rank's step of a 98 304-atom periodic box cut into 8 slabs
(per-layer halo exchange beside the deep halo, one rank at a time on this GPU).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "torchmd-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_MOL, N_ATOMS = 256, 64
# /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA 157.3 TF, dense bf16 MFMA 2.5 PF, HBM 8 TB/s.  The split-bf16 GEMMs
# compute every fp32 product as 6 bf16 MFMA products (exact 3-way split, fp32 accumulation, tn_gemm_sb.hip), so their
# fp32-equivalent ceiling is 2500 / 6 TF.
PEAK = {"mfma_f32_tflops": 157.3, "mfma_bf16_tflops": 2500.0, "hbm_gbs": 8000.0}
SPLIT_PRODUCTS = 6
# HIP kernels behind each profiled class (rocprofv3 names, profiles/r02_kernel_stats*.csv)
KERNEL_OF = {"gemm_edge": "k_gemm_dual_sb2", "gemm_node": "k_gemm_sb1", "message": "k_message_rows8 / k_message_adjoint_rows8",
             "edge_table": "k_edge_interp (+ k_pair_cutoff_hist, k_bucket_scan, k_bucket_scatter)",
             "pair_bwd": "k_embed_gm / k_embed_pair_rb8", "embed_scatter": "k_embed_moments / k_embed_combine", "elementwise": "elementwise",
             "graph": "k_nbr_wave / k_scan_counts"}


# ----------------------------------------------------------------------------------------------- profiling helpers
def profile_begin(model, L, mask=0xFFFFFFFF):
    L.tmdnet_profile_begin(model._engine.handle, mask)


def profile_classes(model, L, stream_ptr, mask):  # kept for tools/*.py
    profile_begin(model, L, mask)
    return L.tmdnet_profile_num_categories()


def profile_collect(model, L, stream_ptr, n):
    ms = (C.c_double * n)()
    fl = (C.c_double * n)()
    by = (C.c_double * n)()
    ln = (C.c_int64 * n)()
    rc = L.tmdnet_profile_end(model._engine.handle, stream_ptr, ms, fl, by, ln)
    assert rc == 0, rc
    names = [L.tmdnet_profile_category_name(i).decode() for i in range(n)]
    return {names[i]: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(ln[i])) for i in range(n)}


def profile_records(model, L, stream_ptr, cap=65536):
    """Per-launch records of the profiled region -> (per-class table, per-kernel groups).  A group = launches that share
    (class, label): one kernel on one shape."""
    cat = (C.c_int32 * cap)()
    ms = (C.c_double * cap)()
    fl = (C.c_double * cap)()
    by = (C.c_double * cap)()
    lab = C.create_string_buffer(64 * cap)
    n = C.c_int64(0)
    rc = L.tmdnet_profile_end_records(model._engine.handle, stream_ptr, cap, cat, ms, fl, by, lab, C.byref(n))
    assert rc == 0, rc
    names = [L.tmdnet_profile_category_name(i).decode() for i in range(L.tmdnet_profile_num_categories())]
    classes = {k: dict(ms=0.0, flops=0.0, bytes=0.0, launches=0) for k in names}
    groups = {}
    for i in range(min(int(n.value), cap)):
        cls = names[cat[i]]
        label = lab.raw[64 * i:64 * i + 64].split(b"\0", 1)[0].decode()
        for d in (classes[cls], groups.setdefault((cls, label), dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))):
            d["ms"] += ms[i]
            d["flops"] += fl[i]
            d["bytes"] += by[i]
            d["launches"] += 1
    return classes, groups


def roofline_of(rec, cls, label, pmc=None, note_kernel=None):
    """rec: summed ms / flops / bytes / launches of one kernel (or class).  A kernel that carries FLOPs is priced against BOTH
    ceilings and reported under the one it sits closer to (the binding one): the K = 128 tensor linears stream their operands once
    and move 32 fp32-FLOP per byte, below the machine balance of the split-bf16 arithmetic (417 TF / 8 TB/s = 52): HBM binds."""
    launches = max(rec["launches"], 1)
    avg_s = rec["ms"] * 1e-3 / launches
    kernel = note_kernel or ("k_tlin9" if label.startswith("tlin9") else KERNEL_OF.get(cls, cls))
    if rec["flops"] > 0:
        ach = rec["flops"] / launches / avg_s / 1e12
        split = not os.environ.get("TMDNET_NO_SPLIT_BF16")
        peak = PEAK["mfma_bf16_tflops"] / SPLIT_PRODUCTS if split else PEAK["mfma_f32_tflops"]
        inst = "v_mfma_f32_32x32x16_bf16 x6 per fp32 product" if split else "v_mfma_f32_32x32x2_f32"
        roof = {"bound": "mfma", "kernel": f"{kernel} [{label}] ({inst})", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak,
                "peak_note": ("fp32-equivalent: dense bf16 MFMA peak 2500 TF / 6 split products; achieved counts algorithmic "
                              "fp32 FLOPs (x6 = executed bf16 FLOPs)") if split else "fp32 MFMA peak",
                "algorithmic_bytes_per_launch": rec["bytes"] / launches}
        hbm = rec["bytes"] / launches / avg_s / 1e9
        if hbm / PEAK["hbm_gbs"] > roof["frac"]:  # the HBM ceiling is the nearer one: report it, keep the matrix-pipe figures beside it
            roof = {"bound": "hbm", "kernel": roof["kernel"], "achieved": hbm, "peak": PEAK["hbm_gbs"], "unit": "GB/s",
                    "frac": hbm / PEAK["hbm_gbs"], "algorithmic_bytes_per_launch": rec["bytes"] / launches,
                    "mfma": {"achieved": roof["achieved"], "peak": roof["peak"], "unit": "TFLOP/s", "frac": roof["frac"],
                             "peak_note": roof["peak_note"]}}
    else:
        ach = rec["bytes"] / launches / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": f"{kernel} [{label}]", "achieved": ach, "peak": PEAK["hbm_gbs"], "unit": "GB/s",
                "frac": ach / PEAK["hbm_gbs"], "algorithmic_bytes_per_launch": rec["bytes"] / launches}
    roof.update({"traffic": None, "launches": rec["launches"], "avg_launch_us": avg_s * 1e6})
    if isinstance(pmc, tuple):  # (bytes, kernel name in the PMC summary, summary): stamped with its provenance
        pmc, pmc_kernel, pmc_file = pmc
        if pmc is not None:
            roof["traffic_source"] = traffic_source(pmc_file, pmc_kernel)
    if pmc is not None:
        roof["traffic"] = pmc
        roof["traffic_ratio"] = pmc / max(rec["bytes"] / launches, 1.0)  # HBM bytes moved / algorithmic bytes
    return roof


def roofline_mfma_of(rec, label, pmc=None):
    """north_star "MFMA utilisation on the linears": the dominant per-atom GEMM (one kernel, one shape).  Its arithmetic is the
    exact 3-way bf16 split: every fp32 product = 6 bf16 MFMA products, so `executed` counts 6 x the algorithmic FLOPs against the
    dense bf16 peak, `achieved` the algorithmic fp32 FLOPs against peak / 6; with K = 128 the launch streams its operands once,
    so the HBM fraction on its algorithmic bytes is reported next to it (the binding roofline of this kernel)."""
    launches = max(rec["launches"], 1)
    avg_s = rec["ms"] * 1e-3 / launches
    split = not os.environ.get("TMDNET_NO_SPLIT_BF16")
    fl, by = rec["flops"] / launches, rec["bytes"] / launches
    peak = PEAK["mfma_bf16_tflops"] / SPLIT_PRODUCTS if split else PEAK["mfma_f32_tflops"]
    kname = "k_tlin9" if label.startswith("tlin9") else "k_gemm_sb1"
    out = {"bound": "mfma", "kernel": f"{kname} [{label}]" + (" (v_mfma_f32_32x32x16_bf16 x6 per fp32 product)" if split else ""),
           "achieved": fl / avg_s / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / peak,
           "executed_bf16_tflops": (SPLIT_PRODUCTS if split else 1) * fl / avg_s / 1e12,
           "executed_frac_of_bf16_peak": (SPLIT_PRODUCTS * fl / avg_s / 1e12 / PEAK["mfma_bf16_tflops"]) if split else None,
           "hbm": {"achieved": by / avg_s / 1e9, "peak": PEAK["hbm_gbs"], "unit": "GB/s", "frac": by / avg_s / 1e9 / PEAK["hbm_gbs"],
                   "algorithmic_bytes_per_launch": by},
           "binding": "hbm" if by / avg_s / 1e9 / PEAK["hbm_gbs"] > fl / avg_s / 1e12 / peak else "mfma",
           "launches": rec["launches"], "avg_launch_us": avg_s * 1e6, "traffic": None}
    if pmc is not None and pmc[0] is not None:
        out["traffic"], out["traffic_ratio"] = pmc[0], pmc[0] / max(by, 1.0)
        out["traffic_source"] = traffic_source(pmc[2], pmc[1])
    return out


def dominant(groups, cls=None):
    items = [(k, v) for k, v in groups.items() if cls is None or k[0] == cls]
    return max(items, key=lambda kv: kv[1]["ms"])


def pmc_kernel_bytes(pmc, cls, label):
    """HBM bytes per launch (PMC, tools/pmc_traffic.py) of the kernel behind a profile label, when the label names one
    template instance whose launches all have this shape (e.g. gemm_dual<2> = k_gemm_dual_sb2<2>: only the 2F -> 3F GEMM)."""
    per = pmc.get("_per_kernel_total", {})
    head = label.split(" ")[0].split("(")[0]
    names = {"gemm_dual<2>": ["k_edge_mlp", "k_gemm_dual_sb2<2>"], "gemm_dual<0>": ["k_gemm_dual_sb2<0>"],
             "launch_message": ["k_message_rows8<8, 4>", "k_message<false>", "k_message<true>", "k_message"],
             "launch_message_adjoint_gd": ["k_message_adjoint_rows8", "k_message_adjoint_gd<false>", "k_message_adjoint_gd<true>", "k_message_adjoint_gd"],
             "launch_edge_tables": ["k_edge_interp<3>", "k_edge_interp<2>", "k_edge_interp<4>", "k_edge_interp<1>"],
             "launch_embed_scatter": ["k_embed_scatter"], "launch_embed_pair_gd": ["k_embed_pair_gd_v4"],
             "launch_embed_combine": ["k_embed_combine<4, 2>"], "launch_embed_gm": ["k_embed_gm<4, 1, 8>"],
             "launch_embed_pair_rb": ["k_embed_pair_rb8<4, 4>"]}.get(head, [])
    if head == "tlin9":  # "tlin9 <variant> 9xNxFxF": the template instance of the variant (tn_tlin9.hip: <prologue, epilogue>)
        names = {"norm": ["k_tlin9<1, 0>"], "update": ["k_tlin9<0, 2>"], "updbwd": ["k_tlin9<2, 0>"], "normbwd": ["k_tlin9<0, 3>"],
                 "normbwd+gate": ["k_tlin9<0, 4>"], "embbwd": ["k_tlin9<0, 5>"], "gate": ["k_tlin9<0, 1>"]}.get(label.split(" ")[1], [])
    for n in names:
        if n in per:
            return per[n], n
    return None, None


def load_pmc_aux(name="pmc_traffic_aux.json"):
    """PMC summary of the auxiliary legs (tools/profile_round.sh: the same two passes over `bench.py --aux-only`; the halo-exchange
    leg: over tools/halo_profile.py -> pmc_traffic_halo.json)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def aux_traffic(prefixes, name="pmc_traffic_aux.json"):
    """(bytes per launch, kernel name, summary) of the first kernel of the aux summary whose name starts with one of `prefixes`
    (the heaviest one when several template instances match), or None."""
    pmc = load_pmc_aux(name)
    per = pmc.get("_per_kernel_total", {})
    for pre in prefixes:
        pre, suf = pre if isinstance(pre, tuple) else (pre, "")  # (prefix, suffix): e.g. the storage mode is the LAST template argument
        hits = [(v, k) for k, v in per.items() if k.startswith(pre) and k.endswith(suf)]
        if hits:
            v, k = max(hits)
            return (v, k, pmc)
    return None


def with_aux_traffic(roof, prefixes, name="pmc_traffic_aux.json"):
    hit = aux_traffic(prefixes, name)
    if hit:
        roof["traffic"] = hit[0]
        roof["traffic_ratio"] = hit[0] / max(roof.get("algorithmic_bytes_per_launch", 0.0), 1.0)
        src = traffic_source(hit[2], hit[1])
        src["file"] = "profiles/" + name
        roof["traffic_source"] = src
    return roof


def traffic_source(pmc, kernel_name):
    """Provenance of a `traffic` figure: it is read from a committed PMC summary (two separate `rocprofv3 --pmc` passes, which
    cannot run inside the timed region), so the line says which file, which commit's library and which kernel it belongs to."""
    meta = pmc.get("_meta", {})
    return {"file": "profiles/pmc_traffic.json", "git_commit": meta.get("git_commit"), "recipe": meta.get("recipe", "tools/profile_round.sh"),
            "kernel_name": kernel_name, "measured_in_this_run": False}


def request_ceiling(kernel_prefix, avg_launch_us):
    """Second ceiling of a gather sweep: its L2 REQUEST rate against the rate the vector memory path sustains for
    random 128-byte pieces (tools/microbench/gather_bw.hip: 6.8 TB/s = 53.1 G requests/s over 256 CUs).  Requests per launch are
    TCC_REQ_sum of the committed counter table (profiles/pmc_issue.json, tools/pmc_issue.py: own rocprofv3 --pmc passes), the
    launch time is this run's.  A sweep at ~0.9 of this ceiling with ~0.4 of the HBM one is bound by the number of requests it
    makes, not by the bytes they carry."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
    except Exception:
        return None
    hits = [(k, v) for k, v in tab.get("kernels", {}).items() if k.startswith(kernel_prefix) and v.get("TCC_REQ_sum")]
    if not hits:
        return None
    name, k = max(hits, key=lambda kv: kv[1]["TCC_REQ_sum"])
    peak = 6.8e12 / 128 / 1e9
    ach = k["TCC_REQ_sum"] / (avg_launch_us * 1e-6) / 1e9
    out = {"bound": "l2_requests", "achieved": ach, "peak": peak, "unit": "G requests/s", "frac": ach / peak,
           "requests_per_launch": k["TCC_REQ_sum"], "read_requests_per_launch": k.get("TCP_TCC_READ_REQ_sum"),
           "peak_note": "random 128-byte pieces, 4 requests in flight per wave at 2 waves per SIMD: 6.8 TB/s (tools/microbench/gather_bw.hip)",
           "source": {"file": "profiles/pmc_issue.json", "kernel_name": name, "git_commit": tab.get("git_commit"), "measured_in_this_run": False}}
    for key in ("frac_wait_any", "frac_wait_inst_any", "frac_active_inst_any", "lds_bank_conflict_frac", "l2_hit_rate"):
        if key in k:
            out[key] = k[key]
    return out


def load_pmc():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path))
    except Exception:
        return {}


# ----------------------------------------------------------------------------------------------- auxiliary legs
def md_latency(model, args_dict, dev, steps=300, dt_fs=1.0):
    """Second half of BASELINE.json's metric: ns/day of single-system stepping (one 64-atom molecule, energies + forces
    every step, neighbour list rebuilt every step) replayed from a captured HIP graph (static_shapes=True, reference
    tensornet.py:277-290 / calculators.py:117-128).  ns/day = steps/s * dt * 0.0864."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    sm = create_model(dict(args_dict, static_shapes=True, max_num_neighbors=64)).to(dev)
    sm.load_state_dict(model.state_dict())
    z, pos, batch = W.synthetic_batch(n_mol=1, n_atoms=N_ATOMS)
    z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)
    replay = sm.capture(z, pos, batch)
    for _ in range(10):
        replay(pos)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        replay(pos)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return {"workload": "1 molecule x 64 atoms, E+F per step, HIP-graph replay", "ms_per_step": dt * 1e3,
            "ns_per_day": 86400.0 / dt * dt_fs * 1e-6, "dt_fs": dt_fs}


AUX_GROUPS = {}  # leg name -> per-kernel groups of its profiled step (written by --breakdown)


def timed_leg(model, L, dev, step, steps, warmup, name=None):
    """warm-up, one fully profiled step (per-kernel groups), `steps` timed steps.  Returns (s per step, classes, groups)."""
    import torch

    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    profile_begin(model, L)
    step()
    classes, groups = profile_records(model, L, sp)
    if name:
        AUX_GROUPS[name] = {f"{c}: {l}": r for (c, l), r in groups.items()}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps, classes, groups, out


def et_c4_leg(dev, L, steps=8, warmup=3, pair_storage="fp32"):
    """BASELINE configs[3]: ET-SPICE.yaml Equivariant Transformer (F=128, 5 layers, 8 heads, K=64, rc=10 A, neighbour
    embedding, distance influence on keys and values, vector cutoff), 256 x 64-atom molecules, E+F.  The config says bf16:
    the GEMMs run on the bf16 matrix pipe through the exact 3-way split, i.e. at fp32 accuracy."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    model = create_model(dict(W.C4_ARGS, pair_storage=pair_storage)).to(dev)
    z, pos, batch = W.synthetic_batch(n_mol=N_MOL, n_atoms=N_ATOMS)
    z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)
    dt, classes, groups, (e, f) = timed_leg(model, L, dev, lambda: model.energy_and_forces(z, pos, batch, None, None, N_MOL), steps, warmup,
                                             name="et_c4" if pair_storage == "fp32" else "et_c4_bf16")
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    (cls, label), rec = dominant(groups)
    return {"workload": "BASELINE configs[3]: ET-SPICE.yaml hyper-parameters, S-mol64 256 x 64 atoms, E+F, random-init (seed 0)",
            "ms_per_step": dt * 1e3, "molecules_per_s": N_MOL / dt,
            "dtype": "f32 (bf16 MFMA, exact 3-way split)" if pair_storage == "fp32" else
                     "f32 arithmetic, per-pair filter rows stored as bf16 (pair_storage='bf16': <= 2e-3 rel. vs the fp32 oracle, tests/test_gpu_et.py)",
            "pairs": model._engine.counts[0],
            "roofline": with_aux_traffic(roofline_of(rec, cls, label, note_kernel=ET_KERNEL_OF.get(cls)),
                                         # the tile sweeps (tn_et_g16.hip) run on this batch; the row sweeps' name for older summaries
                                         [("g16::" + ET_KERNEL_OF.get(cls, "k_et_attn_bwd") + "_g16<", "true>" if pair_storage == "bf16" else "false>"),
                                          (ET_KERNEL_OF.get(cls, "k_et_attn_bwd") + "_p<", "true>" if pair_storage == "bf16" else "false>")]),
            "classes_ms": {k: round(v["ms"], 3) for k, v in classes.items() if v["launches"]}}


ET_KERNEL_OF = {"message": "k_et_attn_fwd", "pair_bwd": "k_et_attn_bwd", "gemm_edge": "k_gemm_dual_sb2", "gemm_node": "k_gemm_sb1"}


def water10k_leg(dev, L, steps=8, warmup=3, dt_fs=1.0):
    """BASELINE configs[4]: 10k-atom periodic water box, O(N) cell-list neighbours, ns/day of E+F stepping on one GPU
    (the AceFF-1.1 checkpoint is not available offline: TensorNet with the C2 hyper-parameters, random-init)."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    model = create_model(dict(W.C2_ARGS, max_num_neighbors=96)).to(dev)
    z, pos, box = W.water_box(n_side=15)  # 10 125 atoms, 46.5 A box (0.1 atoms / A^3)
    z, pos, box = z.to(dev), pos.to(dev), box.to(dev)
    batch = torch.zeros_like(z)
    n = int(z.shape[0])
    dt, classes, groups, (e, f) = timed_leg(model, L, dev, lambda: model.energy_and_forces(z, pos, batch, box, None, 1), steps, warmup,
                                             name="water10k")
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    grid = model.cell_grid(n)
    (cls, label), rec = dominant(groups)
    return {"workload": f"BASELINE configs[4]-like: {n}-atom periodic water box, TensorNet F=128 L=2 K=32 rc=5.0 (random-init), "
                        "cell-list neighbours rebuilt every step, E+F",
            "atoms": n, "pairs": model._engine.counts[0], "cell_grid": grid[:3], "cell_list": bool(grid[3]),
            "ms_per_step": dt * 1e3, "ns_per_day": 86400.0 / dt * dt_fs * 1e-6, "dt_fs": dt_fs,
            "roofline": with_aux_traffic(roofline_of(rec, cls, label), [AUX_KERNEL_PREFIX.get(label.split("(")[0].split(" ")[0], "k_message_adjoint_gd")]),
            "classes_ms": {k: round(v["ms"], 3) for k, v in classes.items() if v["launches"]}}


def one_system_leg(dev, L, steps=5, warmup=2, n_side=32, world=8):
    """SURVEY 8(e), one periodic system over ranks (no counterpart in the reference): per-rank step of a 98 304-atom water box cut
    into 8 slabs, measured on THIS GPU one rank at a time - the deep halo (parallel.SpatialEvaluator, no exchange inside the step)
    and the per-layer halo exchange (parallel.HaloExchangeEvaluator over tmdnet_set_halo_exchange) with a loop-back transport:
    `ms_per_step` = engine step with the gathers and scatters of the 2 L + 1 exchanges + the plan (index arithmetic on the replicated
    positions) + a device-to-device copy of the received rows (a lower bound of the transport), slowest of the sampled ranks (first,
    middle, last); `ms_per_step_with_link_estimate` prices the rows over two xGMI links instead.  Every speed-up here is a PROJECTION."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model
    from torchmdnet_amd.parallel import HaloExchangeEvaluator, SpatialEvaluator

    torch.manual_seed(0)
    args = dict(W.C2_ARGS)
    model = create_model(dict(args)).to(dev)
    z, pos, box = (t.to(dev) for t in W.water_box(n_side=n_side))
    batch = torch.zeros_like(z)

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3

    whole = timed(lambda: model.energy_and_forces(z, pos, batch, box, None, 1, True))

    def loopback(rank, world_, send, send_counts, recv_counts):  # as many rows back as the peers would send
        n = sum(recv_counts)
        return send[:n] if n <= send.shape[0] else send[torch.arange(n, device=send.device) % max(int(send.shape[0]), 1)]

    hx = HaloExchangeEvaluator(lambda zl, pl, bl, wl, ex, grid: model.energy_and_forces(zl, pl, torch.zeros_like(zl), bl, None, 1, True,
                                                                                        atom_weights=wl, halo_exchange=ex, cell_grid=grid),
                               args["cutoff_upper"], transport=loopback)
    deep = SpatialEvaluator(None, args["cutoff_upper"], args["num_layers"])
    ranks = sorted({0, world // 2, world - 1})
    ex_ms, plan_ms, copy_ms, deep_ms, local, owned, active, moved = [], [], [], [], [], [], [], []
    for r in ranks:
        plan = hx.plan(pos, box, r, world)
        t_plan = timed(lambda: hx.plan(pos, box, r, world))
        plan_ms.append(t_plan)
        ex_ms.append(timed(lambda: hx.step(z, pos, box, r, world)) - t_plan)
        local.append(int(plan.gidx.numel())); owned.append(plan.n_own); moved.append(hx.rows_moved * 4)
        active.append(int(model.engine_info("halo_active_rows")))
        # transport term, a LOWER bound measured on this GPU: the rows this rank receives in a step copied device to device once
        # (what a peer-to-peer xGMI write would at least cost at HBM speed; the links themselves are ~7x slower, see `links_ms_estimate`)
        rows = torch.empty(max(hx.rows_moved, 1), dtype=torch.float32, device=dev)
        dst = torch.empty_like(rows)
        copy_ms.append(timed(lambda: dst.copy_(rows)))
        del rows, dst
        gidx, pos_l, box_l, n_own = deep.local_system(pos, box, r, world)
        zl, bl = z[gidx].contiguous(), torch.zeros_like(gidx)
        w = torch.zeros(gidx.numel(), device=dev); w[:n_own] = 1
        deep_ms.append(timed(lambda: model.energy_and_forces(zl, pos_l, bl, box_l, None, 1, True, atom_weights=w)))
    # roofline of the step's dominant kernel (the reverse sweep on the owned rows), middle rank: HIP events of the library's profiler
    r = ranks[len(ranks) // 2]
    profile_begin(model, L)
    hx.step(z, pos, box, r, world)
    classes, groups = profile_records(model, L, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    (cls, label), rec = dominant(groups)
    roof = with_aux_traffic(roofline_of(rec, cls, label, note_kernel="k_message_adjoint_gd<false, true> (owned rows, both halves of owned-ghost pairs)"),
                            ["k_message_adjoint_gd<false, true>", "k_message_adjoint_gd"], "pmc_traffic_halo.json")
    worst = max(range(len(ranks)), key=lambda k: ex_ms[k] + plan_ms[k] + copy_ms[k])
    engine, total = ex_ms[worst], ex_ms[worst] + plan_ms[worst] + copy_ms[worst]
    # two xGMI neighbours per slab, ~153 GB/s per link and direction (MI355X_MICROARCH.md): what the links would add if nothing hid them
    links_ms = max(moved) / 2 / 153e9 * 1e3
    return {"roofline": roof, "classes_ms": {k: round(v["ms"], 3) for k, v in classes.items() if v["launches"]},
            "workload": f"{int(z.shape[0])}-atom periodic water box (TensorNet C2 hyper-parameters, random-init) cut into {world} slabs; one "
                        f"rank at a time on ONE GPU, ranks {ranks} sampled; loop-back transport: a PROJECTION of the multi-GPU step, not a "
                        "measured one (RCCL transport never run on hardware)",
            "atoms": int(z.shape[0]), "ranks": world, "whole_system_one_gpu_ms": whole,
            "ms_per_step": total, "engine_and_row_moves_ms": engine, "plan_ms": plan_ms[worst],
            "transport_lower_bound_ms": copy_ms[worst], "links_ms_estimate": links_ms,
            "ms_per_step_with_link_estimate": engine + plan_ms[worst] + links_ms,
            "projected_speedup_vs_one_gpu": whole / total,
            "projected_speedup_vs_one_gpu_with_link_estimate": whole / (engine + plan_ms[worst] + links_ms),
            "projected_speedup_without_plan_and_links": whole / engine,
            "local_atoms": max(local), "owned_atoms": max(owned), "rows_of_the_per_atom_kernels": max(active),
            "received_bytes_per_rank_and_step": max(moved),
            "deep_halo_ms_per_step": max(deep_ms), "deep_halo_projected_speedup_vs_one_gpu": whole / max(deep_ms)}


AUX_KERNEL_PREFIX = {"launch_message_adjoint_gd": "k_message_adjoint_gd", "launch_message": "k_message<", "gemm": "k_gemm_sb1",
                     "gemm_dual<1>": "k_gemm_dual_sb2<1>", "gemm_dual<2>": "k_gemm_dual_sb2<2>", "tensor_linear": "k_gemm_sb1<0>"}


def tn2_leg(dev, L, steps=8, warmup=3):
    """TensorNet2 + ScalarPlusWeightedCoulomb (the AceFF-2.0 architecture) at the reference test's sizes
    (tests/test_staticshapes.py:9-32: F=128, L=2, K=32, q_dim=16, rc=5), S-mol64 256 x 64 atoms, E+F, random-init."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    args = dict(W.C2_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=16, q_weights=[1.0, 1.0, 1.0])
    model = create_model(dict(args)).to(dev)
    z, pos, batch = W.synthetic_batch(n_mol=N_MOL, n_atoms=N_ATOMS)
    z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)
    dt, classes, groups, (e, f) = timed_leg(model, L, dev, lambda: model.energy_and_forces(z, pos, batch, None, None, N_MOL), steps, warmup,
                                             name="tensornet2")
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    (cls, label), rec = dominant(groups)
    return {"workload": "TensorNet2 + ScalarPlusWeightedCoulomb F=128 L=2 K=32 q_dim=16 rc=5.0 (all-to-all Coulomb), S-mol64 256 x 64 atoms, "
                        "E+F, random-init (seed 0)",
            "ms_per_step": dt * 1e3, "molecules_per_s": N_MOL / dt, "pairs": model._engine.counts[0],
            "roofline": with_aux_traffic(roofline_of(rec, cls, label), [AUX_KERNEL_PREFIX.get(label.split("(")[0].split(" ")[0], "k_gemm_sb1")]),
            "classes_ms": {k: round(v["ms"], 3) for k, v in classes.items() if v["launches"]}}


def training_leg(dev, L, steps=4, warmup=2):
    """Training steps on the bench batch, one SGD step per step, for the three architectures: energy-only (forward +
    tmdnet_energy_param_grads + device-side parameter update) and energy + forces in the DEFAULT mode - the analytic second-order
    pass seeded with d loss / d E (tmdnet_loss_param_grads: one pass delivers the whole gradient of loss(E, F)); for TensorNet the
    two-pass form and the order-2 difference quotient (a test cross-check) are timed beside it.  Not the headline metric: a
    measured number for SURVEY 8(f)4."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    z, pos, batch = W.synthetic_batch(n_mol=N_MOL, n_atoms=N_ATOMS)
    z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)

    def time_step(args, deriv, order, one_pass, q=None, pos_grad=True):
        torch.manual_seed(0)
        model = create_model(dict(args, derivative=deriv)).to(dev)
        model.parameter_gradients = True
        model.force_position_gradient = pos_grad
        model.force_gradient_order = order  # None: the default (analytic second-order pass); 2: central difference, two extra passes
        model.one_pass_training = one_pass  # True (default): the seeded second-order pass delivers the energy term's gradient too
        opt = torch.optim.SGD(model.parameters(), lr=1e-7)

        def step():
            opt.zero_grad()
            y, f = model(z, pos, batch, q=q)
            loss = (y ** 2).mean() + ((f ** 2).mean() if deriv else 0.0)
            loss.backward()
            opt.step()
            return loss

        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize(dev)
        assert torch.isfinite(loss)
        dt = (time.perf_counter() - t0) / steps * 1e3
        del model, opt
        torch.cuda.empty_cache()
        return dt

    out = {"workload": "C2 model, S-mol64 256 x 64 atoms, one optimizer step (SGD) per step, random-init (seed 0)",
           "force_gradient": "analytic second-order pass, one seeded pass for loss(E, F) (the default for every architecture)"}
    out["ms_per_step_energy_only"] = time_step(W.C2_ARGS, False, None, True)
    out["ms_per_step_energy_and_forces"] = time_step(W.C2_ARGS, True, None, True)
    out["energy_and_forces_over_energy_only"] = out["ms_per_step_energy_and_forces"] / out["ms_per_step_energy_only"]
    # the default also delivers d loss / d pos = -H v (pos requires grad in derivative mode, the reference's side effect, so its
    # autograd computes it too); a training loop that never reads pos.grad switches it off with force_position_gradient = False
    out["ms_per_step_energy_and_forces_no_position_gradient"] = time_step(W.C2_ARGS, True, None, True, pos_grad=False)
    out["ms_per_step_energy_and_forces_two_passes"] = time_step(W.C2_ARGS, True, None, False)
    out["ms_per_step_energy_and_forces_difference_quotient"] = time_step(W.C2_ARGS, True, 2, False)
    tn2 = dict(W.C2_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=16, q_weights=[1.0, 1.0, 1.0])
    for name, args, q in (("equivariant_transformer", W.C4_ARGS, None), ("tensornet2", tn2, torch.zeros(N_MOL, device=dev))):
        try:
            r = {"ms_per_step_energy_only": time_step(args, False, None, True, q), "ms_per_step_energy_and_forces": time_step(args, True, None, True, q)}
            r["energy_and_forces_over_energy_only"] = r["ms_per_step_energy_and_forces"] / r["ms_per_step_energy_only"]
        except Exception as exc:  # noqa: BLE001
            r = {"error": repr(exc)}
        out[name] = r
    return out


def cpu_baseline(args_dict, state_dict, budget_s=25.0):
    """Oracle (oracle/tensornet_torch.py: the reference's pure-PyTorch CPU algorithm, autograd forces; the unmodified
    reference cannot travel to the GPU box) on a bounded sample of the same workload, BASELINE.md section 3 protocol:
    chunks of 16 molecules (the [E,3,3,F] temporaries of a 256-molecule batch do not fit a cache), 2 warm-ups, best of 5
    timed evaluations per chunk, chunk times summed.  The thread count is the fastest of {all, 64, 32, 16, 8}."""
    import torch
    from oracle import tensornet_torch as T
    from torchmdnet_amd import workloads as W

    hp = T.hparams_from_args(args_dict)
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    chunk = 16
    z, pos, batch = W.synthetic_batch(n_mol=chunk)
    T.energy_and_forces(sd, hp, z, pos, batch)  # first-touch warm-up
    host_cores = os.cpu_count() or 1
    all_threads = torch.get_num_threads()
    best_t, best = all_threads, None
    for t in sorted({all_threads, 64, 32, 16, 8}):
        if t > all_threads:
            continue
        torch.set_num_threads(t)
        T.energy_and_forces(sd, hp, z, pos, batch)
        t1 = time.perf_counter()
        T.energy_and_forces(sd, hp, z, pos, batch)
        el1 = time.perf_counter() - t1
        if best is None or el1 < best:
            best, best_t = el1, t
    torch.set_num_threads(best_t)
    t_start = time.perf_counter()
    done, total = 0, 0.0
    while done < N_MOL:
        z, pos, batch = W.synthetic_batch(n_mol=chunk, first_seed=done)
        times = []
        for rep in range(7):  # 2 warm-ups + 5 timed
            t1 = time.perf_counter()
            T.energy_and_forces(sd, hp, z, pos, batch)
            if rep >= 2:
                times.append(time.perf_counter() - t1)
        total += min(times)
        done += chunk
        if time.perf_counter() - t_start > budget_s:
            break
    torch.set_num_threads(all_threads)
    ref_note = {}
    try:  # the UNMODIFIED reference and this port timed on the same host (the build container; the reference cannot travel)
        r = json.load(open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json")))
        ref_note = {"reference_same_host_ratio": r["reference_molecules_per_s"] / r["port_molecules_per_s"],
                    "reference_same_host": {k: r[k] for k in ("reference_molecules_per_s", "port_molecules_per_s", "cpu", "threads", "cores") if k in r}}
    except Exception:  # noqa: BLE001
        pass
    return {**ref_note, "value": done / total, "unit": "molecules/s", "cores": host_cores, "threads": best_t, "kind": "port",
            "sample": f"{done} of the {N_MOL} S-mol64 molecules in chunks of {chunk}, oracle/tensornet_torch.py (reference "
                      f"PyTorch CPU algorithm, autograd forces), 2 warm-ups + best of 5 per chunk, {time.perf_counter() - t_start:.1f} s; "
                      "the unmodified reference on the build container: BASELINE.md section 4"}


def gpu_eager_baseline(args_dict, state_dict, dev, chunk=32, reps=3):
    """SURVEY 8(d)'s same-node GPU comparator: the reference ALGORITHM as plain PyTorch-ROCm eager on this MI355X
    (oracle/tensornet_torch.py = reference tensornet.py:54-81 with OPT=False, autograd forces, dense O(n^2) neighbour search per
    molecule), on the same S-mol64 batch, in chunks of `chunk` molecules (the [E, 3, 3, F] temporaries of all 256 at once are a few
    GB: chunked so that the comparator, not the allocator, is timed).  Outside the timed region of the headline; a reported
    baseline like `cpu_baseline`, never the thing measured or shipped."""
    import torch
    from oracle import tensornet_torch as T
    from torchmdnet_amd import workloads as W

    hp = T.hparams_from_args(args_dict)
    sd = {k: v.detach().to(dev) for k, v in state_dict.items()}
    batches = []
    for first in range(0, N_MOL, chunk):
        z, pos, batch = W.synthetic_batch(n_mol=min(chunk, N_MOL - first), n_atoms=N_ATOMS, first_seed=first)
        batches.append((z.to(dev), pos.to(dev), batch.to(dev)))

    def sweep():
        out = []
        for z, pos, batch in batches:
            out.append(T.energy_and_forces(sd, hp, z, pos, batch))
        return out

    sweep()  # warm-up (allocator, kernel selection)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = sweep()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        best = el if best is None or el < best else best
    assert all(torch.isfinite(e).all() and torch.isfinite(f).all() for e, f in res)
    return {"value": N_MOL / best, "unit": "molecules/s", "ms_per_batch": best * 1e3, "kind": "port",
            "what": "oracle/tensornet_torch.py (the reference's pure-PyTorch algorithm, reference tensornet.py:54-81 OPT=False branch, "
                    "autograd forces) as PyTorch-ROCm eager on this GPU",
            "sample": f"all {N_MOL} S-mol64 molecules in chunks of {chunk}, 1 warm-up + best of {reps} sweeps",
            "torch": torch.__version__, "hip": getattr(torch.version, "hip", None), "device": torch.cuda.get_device_name(dev)}


# ----------------------------------------------------------------------------------------------- launch
def self_launch(a):
    """`python bench.py --gpus N` outside torch.distributed: start N ranks of this script, one per GPU."""
    import torch

    have = torch.cuda.device_count()
    if have < a.gpus:
        print(json.dumps({"error": f"--gpus {a.gpus} requested but only {have} GPU(s) are visible; nothing was measured"}))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--rank-step", choices=["auto", "eager", "graph"], default="auto",
                    help="auto: eager two-phase call on one GPU (per-kernel events in the timed region), HIP-graph replay + RCCL "
                         "all-reduce per rank for N > 1; graph: force the N > 1 path (testable on one GPU under torch.distributed.run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-md", action="store_true", help="skip the HIP-graph latency leg (rocprofv3 --pmc cannot trace graph replays)")
    ap.add_argument("--no-aux", action="store_true", help="skip the configs[3] / configs[4] legs (profiling runs)")
    ap.add_argument("--aux-only", action="store_true",
                    help="run ONLY the auxiliary legs (ET, water box, TensorNet2) and print their line: the target of the rocprofv3 passes of tools/profile_round.sh that give these legs their kernel stats and PMC traffic")
    ap.add_argument("--breakdown", type=str, default="", help="write the per-class / per-kernel timing table to this file")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_graph_step = a.rank_step == "graph" or (a.rank_step == "auto" and world > 1)
    use_dist = world > 1 or (use_graph_step and "MASTER_PORT" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_hip(verbose=False)
    if use_dist:
        dist.barrier()
    from torchmdnet_amd import _C, workloads as W
    from torchmdnet_amd.models.model import create_model
    from torchmdnet_amd.parallel import ShardedEvaluator

    L = _C.lib()
    if a.aux_only:  # the profiling target of the auxiliary legs: same legs, same sizes, nothing else in the process
        out = {"aux_only": True}
        for key, leg in (("et_c4", et_c4_leg), ("et_c4_bf16", lambda d, l, **kw: et_c4_leg(d, l, pair_storage="bf16", **kw)),
                         ("water10k", water10k_leg), ("tensornet2", tn2_leg)):
            out[key] = leg(dev, L, steps=a.steps, warmup=a.warmup)
        if a.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(a.breakdown)), exist_ok=True)
            with open(a.breakdown, "w") as fh:
                json.dump({"aux_legs": AUX_GROUPS}, fh, indent=1)
        print(json.dumps(out), flush=True)
        return
    torch.manual_seed(0)
    args_dict = dict(W.C2_ARGS)
    model = create_model(dict(args_dict)).to(dev)
    stream_ptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    # weak scaling: rank r owns molecules [r*256, (r+1)*256) of the synthetic stream
    zw, pw, bw = (t.to(dev) for t in W.synthetic_batch(n_mol=N_MOL, n_atoms=N_ATOMS, first_seed=rank * N_MOL))
    # strong scaling (BASELINE configs[2]): the same 256 molecules on every rank, each evaluates its molecule range
    zs, ps, bs = (t.to(dev) for t in W.synthetic_batch(n_mol=N_MOL, n_atoms=N_ATOMS, first_seed=0))

    if not use_graph_step:
        # one GPU: the eager two-phase call (graph build with its 3-integer read-back, then the enqueue-only E+F call), every
        # launch bracketed by the library's HIP events -> the per-kernel rooflines come from the timed region itself
        def step_weak():
            return model.energy_and_forces(zw, pw, bw, None, None, N_MOL, want_forces=True)

        step_strong = step_weak  # one rank: the shard is the whole batch
        ranges = [(0, N_MOL, 0, N_MOL * N_ATOMS)]
        rank_step = "eager: tmdnet_build_graph (pair-count read-back) + tmdnet_energy_forces"
    else:
        # N > 1: a rank's step is ONE replayed HIP graph (static shapes: neighbour list, forward, reverse pass, zero-padding of
        # the energy vector) + ONE RCCL all-reduce; the shard is cut once outside the timed region (parallel.ShardSession).
        # S-mol64 molecules have 64 atoms, so 64 neighbour slots per atom can never overflow.
        sta = create_model(dict(args_dict, static_shapes=True, max_num_neighbors=N_ATOMS)).to(dev)
        sta.load_state_dict(model.state_dict())
        sharded = ShardedEvaluator(lambda zl, pl, bl, boxl, ql, nm: sta.energy_and_forces(zl, pl, bl, boxl, ql, nm, want_forces=True))
        guard = lambda: (id(sta._engine), sta._engine.generation)  # noqa: E731
        ranges = sharded.plan(bs, N_MOL)  # depends on the batch vector only: computed once, outside the timed region
        ses_strong = sharded.prepare(zs, ps, bs, n_mol=N_MOL, ranges=ranges, graph=True, guard=guard)
        # weak mode keeps the rank's 256 molecules: at that size a replayed graph buys nothing (2.95 vs 2.93 ms, tools/rank_step_probe.py)
        # and the dynamic call takes the embedding in the radial basis (the species count is read back with the pair counts), so
        # the rank step is the N = 1 step + zero-padding + all-reduce on pre-cut tensors and static buffers
        eager = ShardedEvaluator(lambda zl, pl, bl, boxl, ql, nm: model.energy_and_forces(zl, pl, bl, boxl, ql, nm, want_forces=True))
        ses_weak = eager.prepare_local(zw, pw, bw, n_mol=world * N_MOL, mol_lo=rank * N_MOL, atom_lo=rank * N_MOL * N_ATOMS, graph=False)

        def step_weak():
            e, f, _ = ses_weak.step()
            return e, f

        def step_strong():
            e, f, _ = ses_strong.step()
            return e, f

        sta.check_overflow(ranges[rank][3] - ranges[rank][2], ranges[rank][1] - ranges[rank][0])  # replays are unchecked: poll once
        rank_step = ("eager two-phase call on the rank's pre-cut shard + one RCCL all-reduce of the zero-padded energy vector" if a.scaling == "weak"
                     else "HIP-graph replay of the static-shape step (parallel.ShardSession) + one RCCL all-reduce of the energies")

    def timed(step, steps, mask):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        profile_begin(model, L, mask)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        _, groups = profile_records(model, L, stream_ptr)
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, groups, out

    step = step_weak if a.scaling == "weak" else step_strong

    def eager_profile_step():  # per-launch HIP events need eager launches (a replayed graph has no per-kernel events)
        if not use_graph_step:
            return step()
        if a.scaling == "weak":
            return model.energy_and_forces(zw, pw, bw, None, None, N_MOL, want_forces=True)
        m_lo, m_hi, a_lo, a_hi = ranges[rank]
        return model.energy_and_forces(zs[a_lo:a_hi], ps[a_lo:a_hi], bs[a_lo:a_hi] - m_lo, None, None, m_hi - m_lo, want_forces=True)

    for _ in range(max(a.warmup - 1, 0)):
        step()
    # one fully profiled warm-up step picks the dominant kernel class / kernel and records the breakdown
    eager_profile_step()
    torch.cuda.synchronize(dev)
    profile_begin(model, L)
    eager_profile_step()
    classes, groups = profile_records(model, L, stream_ptr)
    # the dominant KERNEL: the (kernel, shape) group with the largest share of the step, whatever its class's total is (two
    # classes within 1 % of each other - per-atom GEMMs and message sweeps - used to flip the choice from run to run)
    (dom_cls, dom_label), _ = dominant(groups, None)
    names = list(classes)
    (_, gemm_label), _ = dominant(groups, "gemm_node")

    el, tgroups, (e, f) = timed(step, a.steps, (1 << names.index(dom_cls)) | (1 << names.index("gemm_node")))
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    mols_per_step = world * N_MOL if a.scaling == "weak" else N_MOL

    other = None
    if world > 1:  # the other scaling mode, same step count, for the record
        ostep = step_strong if a.scaling == "weak" else step_weak
        for _ in range(3):
            ostep()
        oel, _, _ = timed(ostep, a.steps, 0)
        omols = N_MOL if a.scaling == "weak" else world * N_MOL
        other = {"scaling": "strong" if a.scaling == "weak" else "weak", "value": omols * a.steps / oel, "unit": "molecules/s",
                 "ms_per_step": oel / a.steps * 1e3, "molecules_per_step": omols}

    if rank == 0:
        n_pairs, n_edges, _ = model._engine.counts
        pmc = load_pmc()
        # N = 1: records of the timed region.  N > 1: the timed region replays HIP graphs (no per-kernel events), so the
        # records of the eager profiled step of this rank's shard stand in and the line says so.
        live = (dom_cls, dom_label) in tgroups
        rgroups, rsteps = (tgroups, a.steps) if live else (groups, 1)
        kern_rec = rgroups[(dom_cls, dom_label)]
        cls_rec = dict(ms=0.0, flops=0.0, bytes=0.0, launches=0)
        for (c, _), v in rgroups.items():
            if c == dom_cls:
                for k in cls_rec:
                    cls_rec[k] += v[k]
        roof = roofline_of(kern_rec, dom_cls, dom_label, pmc=pmc_kernel_bytes(pmc, dom_cls, dom_label) + (pmc,))
        roof["measured"] = ("HIP events around every launch of this kernel inside the timed region" if live else
                            "HIP events of one eager profiled step of this rank's shard (the timed region replays HIP graphs)")
        roof["launches_per_step"] = kern_rec["launches"] // max(rsteps, 1)
        roof["share_of_step"] = kern_rec["ms"] / rsteps / (el / a.steps * 1e3)
        cavg = roofline_of(cls_rec, dom_cls, "class average", pmc=pmc.get(dom_cls))
        roof["class"] = {"name": dom_cls, "achieved": cavg["achieved"], "frac": cavg["frac"], "launches_per_step":
                         cls_rec["launches"] // max(rsteps, 1), "share_of_step": cls_rec["ms"] / rsteps / (el / a.steps * 1e3),
                         "traffic": cavg.get("traffic"), "traffic_ratio": cavg.get("traffic_ratio")}
        gemm_rec = rgroups.get(("gemm_node", gemm_label))
        if dom_cls == "message":
            rq = request_ceiling("k_message_adjoint_rows8" if "adjoint" in dom_label else "k_message_rows8", roof["avg_launch_us"])
            if rq:
                roof["request_ceiling"] = rq
        out = {
            "metric": "molecules/sec (64-atom molecules) TensorNet E+F",
            "value": mols_per_step * a.steps / el,
            "unit": "molecules/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: TensorNet F=128 L=2 K=32 rc=5.0, S-mol64 256 molecules x 64 atoms per GPU, "
                                    "energies + forces, random-init weights (seed 0)") if a.scaling == "weak" else
                                   ("BASELINE configs[2]: the configs[1] batch (256 molecules x 64 atoms in total) sharded over "
                                    "the GPUs by molecule ranges, energies + forces, random-init weights (seed 0)"),
                       "atoms_per_gpu": N_MOL * N_ATOMS if a.scaling == "weak" else ranges[0][3] - ranges[0][2],
                       "pairs_per_gpu": n_pairs, "directed_edges_per_gpu": n_edges,
                       "parallelism": f"molecule-sharded x{world}, RCCL all-reduce of energies"},
            "roofline": roof,
        }
        out["config"]["rank_step"] = rank_step
        if gemm_rec:
            if gemm_label.startswith("tlin9"):
                gb, gk = pmc_kernel_bytes(pmc, "gemm_node", gemm_label)
                gpmc = (gb, gk, pmc) if gb is not None else None
            else:
                gpmc = ((pmc.get("_per_kernel_total", {}).get("k_gemm_sb1<0>"), "k_gemm_sb1<0>", pmc)
                        if gemm_label.startswith("tensor_linear") else None)
            out["roofline_mfma"] = roofline_mfma_of(gemm_rec, gemm_label, pmc=gpmc)
        if use_dist:
            out["ranks_seen_by_rccl"] = dist.get_world_size()
            out["rccl"] = {"backend": dist.get_backend(), "version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                           "collective_per_step": f"all_reduce(sum) of a zero-padded fp32[{mols_per_step}] energy vector"}
        if other:
            out["other_scaling_mode"] = other
        # north_star: "achieved HBM GB/s on the scatter": the CSR message sweeps, per kernel, from the fully profiled step
        # (HIP events around each launch); algorithmic bytes = every distinct tensor once
        msg = {lab: v for (c, lab), v in groups.items() if c == "message"}
        if msg:
            tot = dict(ms=sum(v["ms"] for v in msg.values()), flops=0.0, bytes=sum(v["bytes"] for v in msg.values()),
                       launches=sum(v["launches"] for v in msg.values()))
            rs = roofline_of(tot, "message", "class average", pmc=pmc.get("message"))
            rs["launches_per_step"] = tot["launches"]
            rs["kernels"] = {}
            for lab, v in msg.items():
                r1 = roofline_of(v, "message", lab, pmc=pmc_kernel_bytes(pmc, "message", lab)[0])
                rs["kernels"][lab.split("(")[0]] = {k: r1[k] for k in ("achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch",
                                                                        "traffic", "traffic_ratio") if k in r1}
            out["roofline_scatter"] = rs
        if world == 1 and not a.no_md:
            try:  # an auxiliary leg must never cost the main line
                out["md_single_system"] = md_latency(model, args_dict, dev)
            except Exception as exc:  # noqa: BLE001
                out["md_single_system"] = {"error": repr(exc)}
        if world == 1 and not a.no_aux:
            for key, leg in (("et_c4", et_c4_leg), ("et_c4_bf16", lambda d, l: et_c4_leg(d, l, pair_storage="bf16")),
                             ("water10k", water10k_leg), ("tensornet2", tn2_leg), ("training_c2", training_leg),
                             ("one_system_8_slabs", one_system_leg)):
                try:
                    out[key] = leg(dev, L)
                except Exception as exc:  # noqa: BLE001
                    out[key] = {"error": repr(exc)}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args_dict, model.state_dict())
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(exc)}
            try:
                out["gpu_eager_baseline"] = gpu_eager_baseline(args_dict, model.state_dict(), dev)
                out["gpu_eager_baseline"]["hip_engine_over_eager"] = out["value"] / out["gpu_eager_baseline"]["value"]
            except Exception as exc:  # noqa: BLE001
                out["gpu_eager_baseline"] = {"error": repr(exc)}
            torch.cuda.empty_cache()
        if a.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(a.breakdown)), exist_ok=True)
            with open(a.breakdown, "w") as fh:
                json.dump({"one_step_profiled_ms": classes, "kernels": {f"{c}: {lab}": v for (c, lab), v in groups.items()},
                           "step_ms": el / a.steps * 1e3, "aux_legs": AUX_GROUPS}, fh, indent=1)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
