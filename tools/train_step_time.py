"""Times of the training step of the C2 model on the bench batch (developer tool; the same leg as bench.py's `training_c2`) and of
the analytic second-order pass alone (tmdnet_force_param_grads): writes gpurun_out/train_step.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
import bench
from torchmdnet_amd import _C, workloads as W
from torchmdnet_amd.models.model import create_model

dev = torch.device("cuda:0")
out = {"training_c2": bench.training_leg(dev, _C.lib())}
torch.manual_seed(0)
model = create_model(dict(W.C2_ARGS, derivative=True)).to(dev)
z, pos, batch = W.synthetic_batch(n_mol=bench.N_MOL, n_atoms=bench.N_ATOMS)
z, pos, batch = z.to(dev), pos.to(dev), batch.to(dev)
v = torch.randn_like(pos)
for _ in range(2):
    model.force_term_parameter_gradients(z, pos, batch, None, None, bench.N_MOL, v)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    model.force_term_parameter_gradients(z, pos, batch, None, None, bench.N_MOL, v)
torch.cuda.synchronize()
out["second_order_pass_ms"] = (time.perf_counter() - t0) / 5 * 1e3
out["second_order_workspace_GiB"] = model._engine.hvp_ws.numel() / 2 ** 30
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "train_step.json"), "w"), indent=1)
print(json.dumps(out))
