"""Stage-by-stage diagnostics of the HIP path against the kernel-level CPU specification
(oracle/tensornet_adjoint.py).  Developer tool for gpurun sessions; prints one line per tensor."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from oracle import tensornet_adjoint as A, tensornet_torch as T
from torchmdnet_amd.models.model import create_model
from torchmdnet_amd import workloads as W

def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)

g = torch.load(os.path.join(ROOT, "tests/golden/tiny_ref.pt"))
args = g["args"]; sd = g["state_dict"]
m = create_model(dict(args)); m.load_state_dict(sd); m = m.cuda()
z, pos, batch, q = g["z"], g["pos"], g["batch"], g["q"]
E, F = m(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())
torch.cuda.synchronize()
print("counts", m._engine.counts)
hp = T.hparams_from_args(args)
Er, Fr, c = A.energy_forces(sd, hp, z, pos, batch, q=q, want_cache=True)
print("graph P ref", c["graph"]["P"], "E ref", c["graph"]["rows"].shape[0])
n, Fh, K = z.shape[0], hp["hidden_channels"], hp["num_rbf"]
P1 = c["graph"]["P"] + 1
for name, shape, ref in [("phi", (P1, K), c["phi"]), ("Q", (P1, 3, Fh), c["Q"]), ("u0", (n, 9, Fh), c["u0"]),
                         ("X_embed", (n, 9, Fh), c["X_embed"]), ("X_layer0", (n, 9, Fh), c["layers"][1]["X"]),
                         ("X_layer1", (n, 9, Fh), c["X_final"]), ("x", (n, Fh), c["x"])]:
    got = m.debug_tensor(name, shape).cpu()
    print(f"{name:10s} rel={rel(got, ref):.3e}  nan={torch.isnan(got).any().item()}")
print("E rel", rel(E.cpu(), g["E"]), "F rel", rel(F.cpu(), g["F"]))
print("E", E.cpu().flatten().tolist(), g["E"].flatten().tolist())
print("F[:3]", F.cpu()[:3].tolist(), g["F"][:3].tolist())

# C2 quick timing
torch.manual_seed(0)
mc = create_model(dict(W.C2_ARGS)).cuda()
zc, pc, bc = W.synthetic_batch(n_mol=256)
zc, pc, bc = zc.cuda(), pc.cuda(), bc.cuda()
for _ in range(3):
    Ec, Fc = mc(zc, pc, bc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    Ec, Fc = mc(zc, pc, bc)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"C2 256x64: {dt*1e3:.3f} ms/step  -> {256/dt:.0f} molecules/s ; counts {mc._engine.counts}; finite {torch.isfinite(Fc).all().item()}")
gc = torch.load(os.path.join(ROOT, "tests/golden/c2_ref.pt"))
print("C2 E rel (first 4 mol)", rel(Ec.cpu()[:4].view(-1,1), gc["E"]), "F rel", rel(Fc.cpu()[:256], gc["F"]))
