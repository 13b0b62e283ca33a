"""Developer A/B for the fused tensor linears (run once per setting of TMDNET_TLIN9_V1 / TMDNET_TLIN9_STAGGER): bit-identical
repeats of the C2 and TensorNet2 bench models, the C2 step time over a few replays, and per-launch times of the tlin9 variants."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
import bench
from torchmdnet_amd import workloads as W, _C
from torchmdnet_amd.models.model import create_model

dev = torch.device("cuda:0")
out = {"env": {k: os.environ.get(k) for k in ("TMDNET_TLIN9_V1", "TMDNET_TLIN9_STAGGER")}}
z, pos, batch = (t.to(dev) for t in W.synthetic_batch(n_mol=256, n_atoms=64))
for name, args, q in (("c2", dict(W.C2_ARGS), None),
                      ("tn2", dict(W.C2_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=16, q_weights=[1.0, 1.0, 1.0]),
                       torch.tensor([float(m % 3 - 1) for m in range(256)], device=dev))):
    torch.manual_seed(0)
    model = create_model(args).to(dev)
    E, F = model.energy_and_forces(z, pos, batch, None, q, 256)
    same = []
    for _ in range(4):
        E2, F2 = model.energy_and_forces(z, pos, batch, None, q, 256)
        same.append(bool(torch.equal(E, E2) and torch.equal(F, F2)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        model.energy_and_forces(z, pos, batch, None, q, 256)
    torch.cuda.synchronize()
    out[name] = {"repeat_bit_identical": same, "ms_per_step": (time.perf_counter() - t0) / 30 * 1e3,
                 "maxdiff_F": float((F - F2).abs().max()), "E0": float(E[0])}
    if name == "c2":
        L = _C.lib()
        dt, classes, groups, _ = bench.timed_leg(model, L, dev, lambda: model.energy_and_forces(z, pos, batch, None, None, 256), 10, 3, name="ab")
        out["c2"]["tlin9_us"] = {lab.split(" 9x")[0]: round(v["ms"] / max(v["launches"], 1) * 1e3, 1) for (c, lab), v in groups.items() if "tlin9" in lab}
        out["c2"]["gemm_node_ms"] = classes["gemm_node"]["ms"]
print(json.dumps(out))
