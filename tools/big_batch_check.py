"""2048 x 64 atoms (131 k atoms, 1.5 M pairs): results identical to a 256-molecule subset, throughput (developer tool)."""
import os, sys, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
m = create_model(dict(W.C2_ARGS)).cuda()
z, pos, batch = W.synthetic_batch(n_mol=2048)
z, pos, batch = z.cuda(), pos.cuda(), batch.cuda()
E, F = m(z, pos, batch)
sel = batch < 256
E2, F2 = m(z[sel], pos[sel], batch[sel])
print("2048 mol: max rel diff vs 256-mol subset E", ((E[:256]-E2).abs().max()/E2.abs().max()).item(), "F", ((F[sel]-F2).abs().max()/F2.abs().max()).item(), "finite", torch.isfinite(F).all().item())
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): m(z, pos, batch)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print(f"{dt*1e3:.2f} ms/step -> {2048/dt:.0f} mol/s", m._engine.counts)
