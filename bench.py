"""bench.py -- molecules/s of the TensorNet energy+force path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one full pass of the hot path over one batch resident in HBM: neighbour graph build,
radial functions, embedding, L interaction layers, readout, per-molecule energies AND the hand-written
reverse pass for the forces (BASELINE configs[1]: TensorNet 128-hidden 2-layer, synthetic 64-atom x
256-molecule batch, fp32).  Multi-GPU = weak scaling: every rank evaluates its own 256 molecules
(no data-path collective) and the per-molecule energies are all-reduced over RCCL each step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events recorded by the library
on the launch stream around every launch of the dominant kernel class during the timed region;
`cpu_baseline` is the oracle (a restatement of the reference's PyTorch CPU path) timed on this host.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "torchmd-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_MOL, N_ATOMS = 256, 64
# /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA 157.3 TF, dense bf16 MFMA 2.5 PF, HBM 8 TB/s.  The pair-row GEMMs
# (class gemm_edge) compute every fp32 product as 6 bf16 MFMA products (exact 3-way split, fp32 accumulation,
# tn_gemm_sb.hip), so their fp32-equivalent ceiling is 2500 / 6 TF; the per-atom GEMMs run on the fp32 MFMA pipe.
PEAK = {"mfma_f32_tflops": 157.3, "mfma_bf16_tflops": 2500.0, "hbm_gbs": 8000.0}
SPLIT_PRODUCTS = 6


def profile_classes(model, L, stream_ptr, mask):
    n = L.tmdnet_profile_num_categories()
    L.tmdnet_profile_begin(model._engine.handle, mask)
    return n


def profile_collect(model, L, stream_ptr, n):
    ms = (C.c_double * n)()
    fl = (C.c_double * n)()
    by = (C.c_double * n)()
    ln = (C.c_int64 * n)()
    rc = L.tmdnet_profile_end(model._engine.handle, stream_ptr, ms, fl, by, ln)
    assert rc == 0, rc
    names = [L.tmdnet_profile_category_name(i).decode() for i in range(n)]
    return {names[i]: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(ln[i])) for i in range(n)}


def md_latency(model, args_dict, dev, steps=300, dt_fs=1.0):
    """Second half of BASELINE.json's metric: ns/day of single-system stepping (one 64-atom molecule,
    energies + forces every step, neighbour list rebuilt every step) replayed from a captured HIP graph
    (static_shapes=True, reference tensornet.py:277-290 / calculators.py:117-128).  ns/day = steps/s * dt * 0.0864."""
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


def cpu_baseline(args_dict, state_dict, budget_s=20.0):
    """Oracle (oracle/tensornet_torch.py: the reference's pure-PyTorch CPU algorithm, autograd forces)
    on a bounded sample of the same workload: batches of 16 molecules until ~budget_s seconds."""
    from oracle import tensornet_torch as T
    from torchmdnet_amd import workloads as W

    hp = T.hparams_from_args(args_dict)
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    chunk = 16
    z, pos, batch = W.synthetic_batch(n_mol=chunk)
    T.energy_and_forces(sd, hp, z, pos, batch)  # warm-up
    # the fastest thread count on this host is the baseline (many-core hosts oversubscribe small tensor ops)
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
    budget_s = max(budget_s - 10 * best, 5.0)
    t0 = time.perf_counter()
    done = 0
    while True:
        z, pos, batch = W.synthetic_batch(n_mol=chunk, first_seed=done % N_MOL)
        t1 = time.perf_counter()
        T.energy_and_forces(sd, hp, z, pos, batch)
        done += chunk
        if time.perf_counter() - t0 > budget_s or done >= N_MOL:
            break
    el = time.perf_counter() - t0
    # input generation is excluded by timing only the evaluations? keep it simple: it is < 1 % of el
    return {"value": done / el, "unit": "molecules/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} of the {N_MOL} S-mol64 molecules in batches of {chunk}, oracle/tensornet_torch.py "
                      f"(reference PyTorch CPU algorithm, autograd forces), {el:.1f} s"}


# HIP kernel behind each profiled class (rocprofv3 names; profiles/r01_kernel_stats.csv)
KERNEL_OF = {"gemm_edge": "k_gemm_dual_sb2", "gemm_node": "k_gemm_sb1", "message": "k_message_tile / k_message_adjoint",
             "pair_bwd": "k_pair_gd_v4 / k_embed_pair_gd_v4"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-md", action="store_true", help="skip the single-system HIP-graph latency leg (profiling runs)")
    ap.add_argument("--breakdown", type=str, default="", help="write the per-class timing table to this file")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_hip(verbose=False)
    if world > 1:
        dist.barrier()
    from torchmdnet_amd import _C, workloads as W
    from torchmdnet_amd.models.model import create_model

    L = _C.lib()
    torch.manual_seed(0)
    args_dict = dict(W.C2_ARGS)
    model = create_model(dict(args_dict)).to(dev)
    # weak scaling: rank r owns molecules [r*256, (r+1)*256) of the synthetic stream
    z, pos, batch = W.synthetic_batch(n_mol=N_MOL, n_atoms=N_ATOMS, first_seed=rank * N_MOL)
    z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)
    e_all = torch.zeros(world * N_MOL, dtype=torch.float32, device=dev)
    stream_ptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        e, f = model.energy_and_forces(z, pos, batch, None, None, N_MOL, want_forces=True)
        if world > 1:
            e_all.zero_()
            e_all[rank * N_MOL:(rank + 1) * N_MOL] = e
            dist.all_reduce(e_all)  # RCCL over xGMI: the path's only exchange
        return e, f

    for _ in range(max(a.warmup - 1, 0)):
        step()
    # one fully profiled warm-up step picks the dominant kernel class and records the breakdown
    step()
    torch.cuda.synchronize(dev)
    ncat = profile_classes(model, L, stream_ptr, 0xFFFFFFFF)
    step()
    table = profile_collect(model, L, stream_ptr, ncat)
    dominant = max(table, key=lambda k: table[k]["ms"])
    dom_idx = list(table).index(dominant)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    profile_classes(model, L, stream_ptr, 1 << dom_idx)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        e, f = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    timed = profile_collect(model, L, stream_ptr, ncat)[dominant]
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    assert torch.isfinite(e).all() and torch.isfinite(f).all()

    if rank == 0:
        n_pairs, n_edges, _ = model._engine.counts
        launches = max(timed["launches"], 1)
        avg_s = timed["ms"] * 1e-3 / launches
        if timed["flops"] > 0:
            ach = timed["flops"] / launches / avg_s / 1e12
            split = dominant == "gemm_edge" and not os.environ.get("TMDNET_NO_SPLIT_BF16")
            peak = PEAK["mfma_bf16_tflops"] / SPLIT_PRODUCTS if split else PEAK["mfma_f32_tflops"]
            inst = "v_mfma_f32_32x32x16_bf16 x6 per fp32 product" if split else "v_mfma_f32_32x32x2_f32"
            roof = {"bound": "mfma", "kernel": f"{dominant} ({KERNEL_OF.get(dominant, dominant)}, {inst})", "achieved": ach,
                    "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "peak_note": ("fp32-equivalent: dense bf16 MFMA peak 2500 TF / 6 split products; achieved counts "
                                  "algorithmic fp32 FLOPs (x6 = executed bf16 FLOPs)") if split else "fp32 MFMA peak"}
        else:
            ach = timed["bytes"] / launches / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": PEAK["hbm_gbs"], "unit": "GB/s",
                    "frac": ach / PEAK["hbm_gbs"]}
        roof.update({"traffic": None, "launches_per_step": launches // max(a.steps, 1), "avg_launch_us": avg_s * 1e6,
                     "share_of_step": timed["ms"] / (el * 1e3)})
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roof["traffic"] = json.load(open(pmc)).get(dominant)
            except Exception:
                pass
        out = {
            "metric": "molecules/sec (64-atom molecules) TensorNet E+F",
            "value": world * N_MOL * a.steps / el,
            "unit": "molecules/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: TensorNet F=128 L=2 K=32 rc=5.0, S-mol64 256 molecules x 64 atoms "
                                   "per GPU, energies + forces, random-init weights (seed 0)",
                       "atoms_per_gpu": N_MOL * N_ATOMS, "pairs_per_gpu": n_pairs, "directed_edges_per_gpu": n_edges,
                       "parallelism": f"molecule-sharded x{world}, RCCL all-reduce of energies"},
            "roofline": roof,
        }
        # north_star: "achieved HBM GB/s on the scatter": the CSR message sweeps (class "message"), from the fully profiled
        # warm-up step (HIP events around each launch); algorithmic bytes as defined in DESIGN.md section 4
        msg = table.get("message")
        if msg and msg["launches"]:
            ach_m = msg["bytes"] / (msg["ms"] * 1e-3) / 1e9
            out["roofline_scatter"] = {"bound": "hbm", "kernel": f"message ({KERNEL_OF['message']})", "achieved": ach_m,
                                       "peak": PEAK["hbm_gbs"], "unit": "GB/s", "frac": ach_m / PEAK["hbm_gbs"],
                                       "launches_per_step": msg["launches"], "avg_launch_us": msg["ms"] * 1e3 / msg["launches"],
                                       "traffic": None}
            try:
                out["roofline_scatter"]["traffic"] = json.load(open(pmc)).get("message")
            except Exception:
                pass
        if world == 1 and not a.no_md:
            try:  # an auxiliary leg must never cost the main line
                out["md_single_system"] = md_latency(model, args_dict, dev)
            except Exception as exc:  # noqa: BLE001
                out["md_single_system"] = {"error": repr(exc)}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args_dict, model.state_dict())
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(exc)}
        if a.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(a.breakdown)), exist_ok=True)
            with open(a.breakdown, "w") as fh:
                json.dump({"one_step_profiled_ms": table, "step_ms": el / a.steps * 1e3}, fh, indent=1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
