"""Developer probe: bit-level repeatability of the C2 forces with individual tlin9 variants on the first-generation kernel."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model
    dev = torch.device("cuda:0")
    z, pos, batch = (t.to(dev) for t in W.synthetic_batch(n_mol=256, n_atoms=64))
    torch.manual_seed(0)
    model = create_model(dict(W.C2_ARGS)).to(dev)
    E, F = model.energy_and_forces(z, pos, batch, None, None, 256)
    diffs = []
    for _ in range(5):
        E2, F2 = model.energy_and_forces(z, pos, batch, None, None, 256)
        diffs.append(int((F != F2).sum()))
    print(json.dumps({"mask": os.environ.get("TMDNET_TLIN9_V1_MASK"), "differing_force_entries": diffs}))
else:
    for mask in ("0", "0x8", "0x10", "0x20", "0x40", "0x78", "0x70", "0x68", "0x58", "0x38"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TMDNET_TLIN9_V1_MASK=mask), capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
