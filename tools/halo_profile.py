"""Developer probe (through gpurun): where one rank's step of the halo exchange goes - engine classes and the largest kernels (the
engine's own profiler), the exchanges (events around the callbacks), the rest (host) - for slab `rank` of `world` of a water box."""
import json, os, sys, time

R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "torchmd-net_amd"))
import torch
import bench as B
from torchmdnet_amd import _C, workloads as W
from torchmdnet_amd.models.model import create_model, _stream_ptr
from torchmdnet_amd.parallel import HaloExchangeEvaluator


def main(n_side=32, world=8, rank=3):
    args = dict(W.C2_ARGS)
    torch.manual_seed(0)
    model = create_model(dict(args)).cuda()
    z, pos, box = (t.cuda() for t in W.water_box(n_side=n_side))
    ex_ms = []

    def loopback(rank_, world_, send, send_counts, recv_counts):  # as many rows back as a peer would send
        n = sum(recv_counts)
        return send[:n] if n <= send.shape[0] else send[torch.arange(n, device=send.device) % max(int(send.shape[0]), 1)]

    hx = HaloExchangeEvaluator(lambda zl, pl, bl, wl, ex, grid: model.energy_and_forces(zl, pl, torch.zeros_like(zl), bl, None, 1, True,
                                                                                        atom_weights=wl, halo_exchange=timed_ex(ex), cell_grid=grid),
                               args["cutoff_upper"], transport=loopback)

    def timed_ex(ex):
        def f(stage, rows, inv):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ex(stage, rows, inv); b.record()
            ex_ms.append((a, b))
        return f

    for _ in range(3):
        hx.step(z, pos, box, rank, world)
    L = _C.lib()
    sp = _stream_ptr(pos.device)
    out = {}
    plan = hx.plan(pos, box, rank, world)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        hx.plan(pos, box, rank, world)
    torch.cuda.synchronize(); out["plan_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    ex_ms.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        hx.step(z, pos, box, rank, world)
    torch.cuda.synchronize(); out["step_ms_wall_with_plan"] = (time.perf_counter() - t0) / 5 * 1e3
    out["exchange_ms_per_step"] = sum(a.elapsed_time(b) for a, b in ex_ms) / 5
    B.profile_begin(model, L)
    hx.step(z, pos, box, rank, world)
    classes, groups = B.profile_records(model, L, sp)
    out["engine_classes_ms"] = {k: round(v["ms"], 3) for k, v in classes.items()}
    out["engine_ms"] = round(sum(v["ms"] for v in classes.values()), 3)
    out["largest_kernels"] = sorted(((round(v["ms"], 3), v["launches"], f"{c}: {l}") for (c, l), v in groups.items()), reverse=True)[:14]
    out["local_atoms"], out["owned"], out["active_rows"] = int(plan.gidx.numel()), plan.n_own, int(model.engine_info("halo_active_rows"))
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(R, "gpurun_out", "halo_profile.json"), "w"), indent=1)


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
