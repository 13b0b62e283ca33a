"""Large periodic water box through the cell list: finite, zero net force, translation invariance, ns/day, peak memory
(developer tool).  N_SIDE=32: 98 304 atoms; N_SIDE=70 RECOMPUTE=1: 1 029 000 atoms on one 288 GB device (option
"recompute_pair_rows": no per-pair rows in the workspace)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
m = create_model(dict(W.C2_ARGS, max_num_neighbors=96)).cuda()
z, pos, box = W.water_box(n_side=int(os.environ.get("N_SIDE", "32")))
z, pos, box = z.cuda(), pos.cuda(), box.cuda()
batch = torch.zeros_like(z)
if os.environ.get("RECOMPUTE"):
    m.set_engine_option("recompute_pair_rows", 1)
E, F = m(z, pos, batch, box=box)
print("atoms", z.numel(), "pairs/edges", m._engine.counts[:2], "grid", m.cell_grid(z.numel()))
print("finite", torch.isfinite(E).all().item(), torch.isfinite(F).all().item(), "E", E.item(), "|sum F| / max|F|",
      (F.sum(0).abs().max() / F.abs().max()).item())
E2, F2 = m(z, pos + torch.tensor([1.234, -2.5, 7.7], device="cuda"), batch, box=box)
print("translation: dE/E", ((E2 - E).abs() / E.abs()).item(), "dF", ((F2 - F).abs().max() / F.abs().max()).item())
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = int(os.environ.get("REPS", "5"))
for _ in range(reps): m(z, pos, batch, box=box)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"{dt*1e3:.2f} ms/step -> {86.4/dt*1e-3:.2f} ns/day at 1 fs; peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
