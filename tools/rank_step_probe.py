"""Developer probe behind DESIGN.md's predicted multi-GPU curves: per-rank E+F step of the S-mol64 batch at the shard sizes of
BASELINE configs[2] (256 / 128 / 64 / 32 molecules = 1, 2, 4, 8 ranks), three ways on ONE GPU:
  eager   - tmdnet_build_graph (reads the pair count back) + tmdnet_energy_forces, as bench.py at N = 1
  static  - static shapes (capacity-sized launches, no read-back), eager launches
  graph   - the static step captured into one HIP graph (parallel.ShardSession, what a rank replays per step)
Writes gpurun_out/rank_step_probe.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
from torchmdnet_amd.parallel import ShardedEvaluator


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
dyn = create_model(dict(W.C2_ARGS)).cuda()
out = {}
for cap in (64, 128):
    sta = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=cap)).cuda()
    sta.load_state_dict(dyn.state_dict())
    for n_mol in (256, 128, 64, 32):
        z, pos, batch = (t.cuda() for t in W.synthetic_batch(n_mol=n_mol))
        n = 40 if n_mol >= 128 else 100
        row = {}
        if cap == 64:
            row["eager_ms"] = timeit(lambda: dyn.energy_and_forces(z, pos, batch, None, None, n_mol), n)
        row["static_ms"] = timeit(lambda: sta.energy_and_forces(z, pos, batch, None, None, n_mol), n)
        ev = ShardedEvaluator(lambda zl, pl, bl, boxl, ql, nm: sta.energy_and_forces(zl, pl, bl, boxl, ql, nm, want_forces=True))
        ses = ev.prepare(z, pos, batch, n_mol=n_mol, graph=True)
        row["graph_ms"] = timeit(lambda: ses.step(), n)
        E, F = dyn.energy_and_forces(z, pos, batch, None, None, n_mol)
        Es, Fs, _ = ses.step()
        row["rel_diff_F"] = ((Fs - F).abs().max() / F.abs().max()).item()
        out[f"cap{cap}_mol{n_mol}"] = row
        print(cap, n_mol, row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rank_step_probe.json"), "w"), indent=1)
