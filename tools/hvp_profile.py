"""The analytic second-order pass alone on the bench batch, a few repetitions (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = create_model(dict(W.C2_ARGS, derivative=True)).to(dev)
z, pos, batch = W.synthetic_batch(n_mol=256, n_atoms=64)
z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)
v = torch.randn_like(pos)
ge = torch.randn(256, device=dev)
for _ in range(int(os.environ.get("REPS", "4"))):  # the seeded pass of one-pass training (with H v), as a training step runs it
    model.force_term_parameter_gradients(z, pos, batch, None, None, 256, v, want_hv=True, ge=ge)
torch.cuda.synchronize()
