"""Single-system latency (ns/day) probe: one 64-atom molecule, E+F per step (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
for nm, na in [(1, 64), (1, 1000), (8, 64)]:
    m = create_model(dict(W.C2_ARGS)).cuda()
    if na == 64:
        z, pos, batch = W.synthetic_batch(n_mol=nm, n_atoms=na)
    else:
        z, pos, box = W.water_box(n_side=7); batch = torch.zeros(z.shape[0], dtype=torch.long)
    z, pos, batch = z.cuda(), pos.cuda(), batch.cuda()
    for _ in range(5): m(z, pos, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): E, F = m(z, pos, batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ms = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=64)).cuda()
    ms.load_state_dict(m.state_dict())
    replay = ms.capture(z, pos, batch)
    for _ in range(5): replay(pos)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): Eg, Fg = replay(pos)
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / n
    err = (Fg - F).abs().max().item() / F.abs().max().item()
    print(f"   HIP graph replay: {dtg*1e3:.3f} ms/step -> {86400/dtg*1e-6:.1f} ns/day at dt=1 fs ; F rel diff vs eager {err:.1e}")
    print(f"{nm} x {z.shape[0]//nm} atoms: {dt*1e3:.3f} ms/step -> {86400/dt*1e-6*1.0:.1f} ns/day at dt=1 fs ; counts {m._engine.counts}")
