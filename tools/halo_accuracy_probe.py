"""Developer probe (through gpurun): where the halo exchange's forces differ from the whole system's, against two noise floors of the
whole system itself - atoms renumbered (summation order) and the box translated and re-wrapped (fp32 coordinates)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "torchmd-net_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
import test_gpu_halo as T
args = dict(W.C2_ARGS)
for n_side in (15, 32):
    z, pos, box = (t.cuda() for t in W.water_box(n_side=n_side))
    torch.manual_seed(0)
    whole = create_model(dict(args)).cuda()
    Ew, Fw = whole.energy_and_forces(z, pos, torch.zeros_like(z), box, None, 1, True)
    # the whole system evaluated twice with atoms renumbered: the summation-order noise floor
    perm = torch.randperm(z.shape[0], device="cuda")
    Ep, Fp = whole.energy_and_forces(z[perm], pos[perm], torch.zeros_like(z), box, None, 1, True)
    Fp2 = torch.empty_like(Fp); Fp2[perm] = Fp
    fmax = Fw.abs().max().item()
    sh = torch.tensor([float(box[0, 0]) * 0.37, 0.0, 0.0], device="cuda")
    Et, Ft = whole.energy_and_forces(z, torch.remainder(pos + sh, torch.diagonal(box)), torch.zeros_like(z), box, None, 1, True)
    print(n_side, "noise floor (whole system translated by 0.37 L along x and wrapped):", ((Ft - Fw).abs().max() / Fw.abs().max()).item())
    print(n_side, "noise floor (renumbered whole system):", ((Fp2 - Fw).abs().max() / fmax).item(), "fmax", fmax)
    E, F, n_local, moved, active = T._threaded(args, 0, z, pos, box, 2)
    err = (F - Fw).abs().max(dim=1).values / fmax
    La = float(box[0, 0]); w = La / 2
    x = torch.remainder(pos[:, 0], La)
    dist_face = torch.minimum(torch.remainder(x, w), w - torch.remainder(x, w))
    print(n_side, "halo max err", err.max().item(), "active", active)
    for lo, hi in ((0, 1), (1, 3), (3, 6), (6, 12), (12, 100)):
        m = (dist_face >= lo) & (dist_face < hi)
        if not bool(m.any()): continue
        print(f"   atoms {lo}-{hi} A from a slab face: n={int(m.sum())} max err {err[m].max().item():.2e} mean {err[m].mean().item():.2e}")
    k = int(err.argmax()); print("   worst atom", k, "F", Fw[k].tolist(), "dF", (F - Fw)[k].tolist(), "dist_face", dist_face[k].item())
