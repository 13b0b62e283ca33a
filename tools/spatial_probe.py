"""Developer probe (through gpurun): one periodic water box decomposed into R slabs (parallel.SpatialEvaluator), every slab's local
system timed on ONE GPU -> the per-rank step time a node of R GPUs would see (max over slabs) against the whole system on one GPU.
The collective (one all-reduce of 3N + 1 floats) is not in these numbers.  Beside the deep halo, the per-layer halo exchange
(parallel.HaloExchangeEvaluator): the same slabs with ONE cutoff of ghosts, the 2 L + 1 exchanges served by a loop-back transport that
hands every rank rows of its own of the right count - the gathers and scatters of the exchange are timed, the links are not
(`halo_bytes_per_step` is what a rank receives over xGMI per step)."""
import json, os, sys, time

R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
from torchmdnet_amd.parallel import HaloExchangeEvaluator, SpatialEvaluator


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    args = dict(W.C2_ARGS)
    torch.manual_seed(0)
    model = create_model(dict(args)).cuda()
    out = {}
    for n_side in (15, 24, 32):
        z, pos, box = (t.cuda() for t in W.water_box(n_side=n_side))
        batch = torch.zeros_like(z)
        whole = timed(lambda: model.energy_and_forces(z, pos, batch, box, None, 1, True))
        rec = {"atoms": int(z.shape[0]), "box_A": float(box[0, 0]), "whole_ms": whole, "ranks": {}}
        ev = SpatialEvaluator(None, args["cutoff_upper"], args["num_layers"])
        for world in (2, 4, 8):
            if ev.halo > float(box[0, 0]):
                continue
            per = []
            for r in range(world):
                gidx, pos_l, box_l, n_own = ev.local_system(pos, box, r, world)
                zl, bl = z[gidx].contiguous(), torch.zeros_like(gidx)
                w = torch.zeros(gidx.numel(), device="cuda"); w[:n_own] = 1
                per.append((int(gidx.numel()), timed(lambda: model.energy_and_forces(zl, pos_l, bl, box_l, None, 1, True, atom_weights=w))))
            rec["ranks"][world] = {"local_atoms_max": max(p[0] for p in per), "step_ms_max": max(p[1] for p in per),
                                   "speedup_vs_one_gpu": whole / max(p[1] for p in per)}
            # per-layer halo exchange: one cutoff of ghosts, loop-back rows
            def loopback(rank, world_, send, send_counts, recv_counts):  # as many rows back as a peer would send
                n = sum(recv_counts)
                return send[:n] if n <= send.shape[0] else send[torch.arange(n, device=send.device) % max(int(send.shape[0]), 1)]

            hx = HaloExchangeEvaluator(lambda zl, pl, bl, wl, ex, grid: model.energy_and_forces(zl, pl, torch.zeros_like(zl), bl, None, 1,
                                                                                                True, atom_weights=wl, halo_exchange=ex,
                                                                                                cell_grid=grid),
                                       args["cutoff_upper"], transport=loopback)
            per = []
            for r in range(world):
                plan = hx.plan(pos, box, r, world)
                t = timed(lambda: hx.step(z, pos, box, r, world))
                t_plan = timed(lambda: hx.plan(pos, box, r, world), reps=3)
                per.append((int(plan.gidx.numel()), t - t_plan, plan.n_ghost, hx.rows_moved * 4, int(model.engine_info("halo_active_rows")),
                            plan.n_own))
            rec["ranks"][world].update({"exchange_local_atoms_max": max(p[0] for p in per), "exchange_step_ms_max": max(p[1] for p in per),
                                        "exchange_ghosts_max": max(p[2] for p in per), "halo_bytes_per_step": max(p[3] for p in per),
                                        "exchange_speedup_vs_one_gpu": whole / max(p[1] for p in per),
                                        "per_atom_kernels_on_owned_rows_only": [p[4] == p[5] for p in per]})
        out[f"water_{n_side}"] = rec
        print(json.dumps({f"water_{n_side}": rec}))
    os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(R, "gpurun_out", "spatial_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
