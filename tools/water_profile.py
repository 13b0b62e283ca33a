"""Developer probe (through gpurun / rocprofv3): a few E+F steps of the 10 125-atom periodic water box (BASELINE configs[4]-like)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
n_side = int(os.environ.get("N_SIDE", 15))
m = create_model(dict(W.C2_ARGS, max_num_neighbors=96)).cuda()
z, pos, box = (t.cuda() for t in W.water_box(n_side=n_side))
batch = torch.zeros_like(z)
for _ in range(int(os.environ.get("STEPS", 4))):
    E, F = m.energy_and_forces(z, pos, batch, box, None, 1)
torch.cuda.synchronize()
print(int(z.shape[0]), m._engine.counts[:2], bool(torch.isfinite(F).all()))
