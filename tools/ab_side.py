"""Same-process A/B of the table side stream (tn_model.h: `side_tab`): the per-pair rows from the radial tables written on a second
stream beside the node kernels, against the single-stream schedule (TMDNET_NO_TABLE_SIDE_STREAM=1, read when the handle is
created).  Two models with identical weights per workload, timed alternately (3 rounds), outputs compared bit for bit.
Developer tool, run through gpurun; prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch  # noqa: E402
from torchmdnet_amd import workloads as W  # noqa: E402
from torchmdnet_amd.models.model import create_model  # noqa: E402

dev = torch.device("cuda", 0)


def make(args, off, **kw):
    if off:
        os.environ["TMDNET_NO_TABLE_SIDE_STREAM"] = "1"
    else:
        os.environ.pop("TMDNET_NO_TABLE_SIDE_STREAM", None)
    torch.manual_seed(0)
    m = create_model(dict(args)).to(dev)
    for k, v in kw.items():
        setattr(m, k, v)
    return m


def timed(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3, out


def leg(name, args, inputs, n_mol, steps=20, **kw):
    z, pos, batch, box = inputs
    on, off = make(args, False, **kw), make(args, True, **kw)
    off.load_state_dict(on.state_dict())
    f_on = lambda: on.energy_and_forces(z, pos, batch, box, None, n_mol)  # noqa: E731
    f_off = lambda: off.energy_and_forces(z, pos, batch, box, None, n_mol)  # noqa: E731
    t_on, t_off = [], []
    for _ in range(3):
        a, (e1, f1) = timed(f_on, steps)
        b, (e0, f0) = timed(f_off, steps)
        t_on.append(a)
        t_off.append(b)
    same = bool(torch.equal(e1, e0) and torch.equal(f1, f0))
    rep = [tuple(t.clone() for t in f_on()) for _ in range(3)]
    stable = all(torch.equal(rep[0][0], r[0]) and torch.equal(rep[0][1], r[1]) for r in rep[1:])
    return {"side_stream_ms": [round(t, 4) for t in t_on], "single_stream_ms": [round(t, 4) for t in t_off],
            "best_side_over_single": round(min(t_on) / min(t_off), 4), "bit_identical_to_single_stream": same,
            "bit_identical_repeats": bool(stable)}


out = {}
z, pos, batch = (t.to(dev) for t in W.synthetic_batch(n_mol=256, n_atoms=64))
out["c2"] = leg("c2", W.C2_ARGS, (z, pos, batch, None), 256, steps=30)
out["et_c4"] = leg("et_c4", W.C4_ARGS, (z, pos, batch, None), 256, steps=8)
try:
    out["et_c4_bf16"] = leg("et_c4_bf16", W.C4_ARGS, (z, pos, batch, None), 256, steps=8, pair_storage="bf16")
except Exception as exc:  # noqa: BLE001
    out["et_c4_bf16"] = {"error": repr(exc)}
zw, pw, bw = (t.to(dev) for t in W.water_box(n_side=15))
out["water10k"] = leg("water10k", dict(W.C2_ARGS, max_num_neighbors=96), (zw, pw, torch.zeros_like(zw), bw), 1, steps=20)
print(json.dumps(out))
