"""C2 batch: eager dynamic-shape step vs static shapes + HIP-graph replay (developer tool)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
m = create_model(dict(W.C2_ARGS)).cuda()
z, pos, batch = (t.cuda() for t in W.synthetic_batch(n_mol=256))
def timeit(f, n=40):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager dynamic     ", round(timeit(lambda: m.energy_and_forces(z, pos, batch, None, None, 256)), 4), "ms")
for mnn in (64, 40):
    sm = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=mnn)).cuda(); sm.load_state_dict(m.state_dict())
    print(f"eager static mnn={mnn}", round(timeit(lambda: sm.energy_and_forces(z, pos, batch, None, None, 256)), 4), "ms")
    r = sm.capture(z, pos, batch, num_systems=256)
    print(f"graph replay mnn={mnn}", round(timeit(lambda: r(pos)), 4), "ms")
    E0, F0 = m(z, pos, batch); E1, F1 = r(pos)
    print("   max rel diff E", ((E1 - E0).abs().max() / E0.abs().max()).item(), "F", ((F1 - F0).abs().max() / F0.abs().max()).item())
