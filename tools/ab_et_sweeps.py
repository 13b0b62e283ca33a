"""A/B of the Equivariant Transformer attention sweeps under developer switches read at the first launch (one setting per
process): TMDNET_ET_NO_G16=1 (one-channel-per-lane kernels of tn_et.hip), TMDNET_ET_G16_SYNC=<steps between barriers, 0 = none>.
usage: TMDNET_ET_G16_SYNC=8 python tools/ab_et_sweeps.py ; prints the step time, the class table and a checksum of the forces."""
import ctypes as C, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import _C, workloads as W
from torchmdnet_amd.models.model import create_model
import bench
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("TMDNET_ET_")}}
if os.environ.get("RAGGED"):  # molecules of 10 .. 64 atoms (uniform), about 16 384 atoms in all: a SPICE-like batch
    import numpy as np
    rng = np.random.default_rng(0)
    zs, ps, bs, tot, m = [], [], [], 0, 0
    while tot < 16384:
        n = int(rng.integers(10, 65))
        zz, pp, _ = W.synthetic_batch(n_mol=1, n_atoms=n, first_seed=1000 + m)
        zs.append(zz); ps.append(pp); bs.append(torch.full((n,), m, dtype=torch.long)); tot += n; m += 1
    z, pos, batch = torch.cat(zs).cuda(), torch.cat(ps).cuda(), torch.cat(bs).cuda()
    N_MOL = m
    out["ragged"] = {"molecules": m, "atoms": tot}
else:
    z, pos, batch = (t.cuda() for t in W.synthetic_batch(n_mol=256))
    N_MOL = 256
L = _C.lib()
for storage in ("fp32", "bf16"):
    torch.manual_seed(0)
    model = create_model(dict(W.C4_ARGS)).to("cuda")
    model.pair_storage = storage
    step = lambda: model.energy_and_forces(z, pos, batch, None, None, N_MOL)
    for _ in range(3): step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(8): e, f = step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ncat = bench.profile_classes(model, L, sp, 0xFFFFFFFF)
    step()
    table = bench.profile_collect(model, L, sp, ncat)
    out[storage] = {"ms_per_step": round(best, 4), "pair_bwd_ms": round(table["pair_bwd"]["ms"], 3), "message_ms": round(table["message"]["ms"], 3),
                    "forces_sha": hashlib.sha1(f.cpu().numpy().tobytes()).hexdigest()[:12], "energy_sha": hashlib.sha1(e.cpu().numpy().tobytes()).hexdigest()[:12]}
# optional: compare with / leave behind the forces of another setting (REF_NPY: read when it exists, written otherwise)
ref = os.environ.get("REF_NPY")
if ref:
    import numpy as np
    fa = f.cpu().numpy()
    if os.path.exists(ref):
        fr = np.load(ref)
        out["vs_ref"] = {"max_abs_diff": float(np.abs(fa - fr).max()), "max_abs": float(np.abs(fr).max()), "n_differ": int((fa != fr).sum()), "n": int(fa.size)}
    else:
        np.save(ref, fa)
print(json.dumps(out))
