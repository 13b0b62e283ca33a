"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/et_*.pt from the UNMODIFIED reference (Equivariant Transformer).

Run in the build container (needs /root/reference):   python oracle/make_golden_et.py

Fixtures
  expected_et_scalar.pt   the reference's own golden vector tests/expected.pkl["equivariant-transformer"]["Scalar"]
                          + the inputs its recipe generates (tests/test_model.py:282-329, tests/utils.py:11-46)
  et_tiny_ref.pt          ET_TINY config (F=32, L=2, H=4, K=16, rc=5, no vector cutoff), 3 ragged molecules: weights,
                          E/F (fp32 and fp64 weights-cast), x/vec after every layer
  et_tiny_vc_ref.pt       same sizes with vector_cutoff=True and distance_influence="keys" (another branch set)
  et_c4_ref.pt            BASELINE configs[3] model (examples/ET-SPICE.yaml hyper-parameters, seed 0), 3 molecules of
                          S-mol64: E/F only
"""
import os
import pickle
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))

from oracle import ref_shims as R  # noqa: E402
from oracle.make_golden import run_ref, sd_checksum  # noqa: E402
from torchmdnet_amd import workloads as W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ragged(seed0, sizes):
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(seed0 + m, n_atoms=n)
        zs.append(torch.from_numpy(zz) % 19 + 1)
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    return torch.cat(zs), torch.cat(ps), torch.cat(bs)


def main():
    os.makedirs(OUT, exist_ok=True)
    mm = R.reference_model_module()
    warnings.simplefilter("ignore")

    # ---- 1. the reference's own golden vector
    R.seed_everything(1234)
    args = R.load_example_args("equivariant-transformer", remove_prior=True, output_model="Scalar", derivative=True)
    model = mm.create_model(args)
    z, pos, batch = R.create_example_batch(n_atoms=5)
    y, f = run_ref(model, z, pos, batch)
    with open(os.path.join(R.REFERENCE_ROOT, "tests", "expected.pkl"), "rb") as fh:
        exp = pickle.load(fh)["equivariant-transformer"]["Scalar"]
    assert (y - exp["pred"]).abs().max() < 1e-5 and (f - exp["deriv"]).abs().max() < 1e-5, ((y - exp["pred"]).abs().max(), (f - exp["deriv"]).abs().max())
    torch.save(
        dict(args=args, z=z, pos=pos.detach(), batch=batch, pred=exp["pred"].detach(), deriv=exp["deriv"].detach(),
             sd_checksum=sd_checksum(model.state_dict())),
        os.path.join(OUT, "expected_et_scalar.pt"),
    )

    # ---- 2./3. tiny configs, ragged molecules, intermediates
    for name, targs, seed in (("et_tiny_ref.pt", dict(W.ET_TINY_ARGS), 7),
                              ("et_tiny_vc_ref.pt", dict(W.ET_TINY_ARGS, vector_cutoff=True, distance_influence="keys"), 8)):
        torch.manual_seed(seed)
        model = mm.create_model(dict(targs))
        z, pos, batch = ragged(100, [7, 12, 20])
        store = {}
        rm = model.representation_model
        hooks = [layer.register_forward_hook(lambda m, i, o, l=l: store.__setitem__(f"layer{l}", (o[0].detach(), o[1].detach())))
                 for l, layer in enumerate(rm.attention_layers)]
        hooks.append(rm.register_forward_hook(lambda m, i, o: store.__setitem__("out", (o[0].detach(), o[1].detach()))))
        y, f = run_ref(model, z, pos, batch)
        for h in hooks:
            h.remove()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        m64 = mm.create_model(dict(targs, precision=64))
        m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
        y64, f64 = run_ref(m64, z, pos.double(), batch)
        inter = {"x_out": store["out"][0], "vec_out": store["out"][1]}
        for l in range(len(rm.attention_layers)):
            inter[f"dx_layer{l}"], inter[f"dvec_layer{l}"] = store[f"layer{l}"]
        torch.save(dict(args=targs, state_dict=sd, z=z, pos=pos, batch=batch, E=y, F=f, E64=y64, F64=f64, inter=inter),
                   os.path.join(OUT, name))

    # ---- 4. C4 model (seed 0) on 3 molecules of S-mol64
    torch.manual_seed(0)
    cmodel = mm.create_model(dict(W.C4_ARGS))
    z, pos, batch = W.synthetic_batch(n_mol=3)
    y, f = run_ref(cmodel, z, pos, batch)
    torch.save(dict(args=dict(W.C4_ARGS), n_mol=3, E=y, F=f, sd_checksum=sd_checksum(cmodel.state_dict())),
               os.path.join(OUT, "et_c4_ref.pt"))
    for fn in sorted(os.listdir(OUT)):
        if "et" in fn:
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
