"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/second_order_ref.pt from the UNMODIFIED reference: what its second autograd
pass produces for a force-matching loss (torchmdnet/models/model.py:618-628: in training mode the forces are taken with
create_graph=True, so a loss on them back-propagates through the first derivative).

For every fixture below the reference model is rebuilt in fp64 from the fixture's weights, put in training mode, evaluated on the
fixture's inputs, and  s = v . d(sum_m E_m)/d pos = - sum_i v_i . F_i  is back-propagated (v seeded).  Stored per fixture:
v, s, H v = d s / d pos and d s / d theta for every parameter (by state-dict key), in fp64; and, for a loss of energies and forces
loss = sum_m ge_m E_m + sum_i v_i . F_i (ge seeded), its gradient in every parameter and in the positions from one backward of the
reference (what one-pass training must reproduce: minus the gradient of S = s - sum_m ge_m E_m).

tests/test_oracle.py pins the three specifications oracle/{tensornet,et,tn2}_second_order.py to these numbers (the -m gpu tests
compare the engine with the specifications).

Run in the build container (needs /root/reference):   python oracle/make_golden_second_order.py
"""
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))

from oracle import ref_shims as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"]
V_SEED = 41


def direction(shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(V_SEED), dtype=torch.float64)


def second_order(mm, g):
    args = dict(g["args"], precision=64, derivative=True)
    model = mm.create_model(args)
    model.load_state_dict({k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()})
    model.train()  # forces with create_graph=True (model.py:618-628); none of these models has dropout or batch statistics
    pos = g["pos"].double().clone().requires_grad_(True)
    q = g.get("q")
    box = g.get("box")
    y, f = model(g["z"], pos, g["batch"], box=None if box is None else box.double(), q=None if q is None else q.double())
    v = direction(pos.shape)
    s = -(v * f).sum()
    s.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    out = dict(v=v, s=s.detach(), Hv=pos.grad.detach().clone(), E=y.detach(), F=f.detach(), grads=grads)
    # a loss of energies AND forces, loss = sum_m ge_m E_m + sum_i v_i . F_i: its whole gradient from ONE backward of the reference
    model.zero_grad()
    pos2 = g["pos"].double().clone().requires_grad_(True)
    y2, f2 = model(g["z"], pos2, g["batch"], box=None if box is None else box.double(), q=None if q is None else q.double())
    ge = torch.randn(y2.shape[0], generator=torch.Generator().manual_seed(V_SEED + 1), dtype=torch.float64)
    ((ge * y2.view(-1)).sum() + (v * f2).sum()).backward()
    out.update(ge=ge, loss_grads={k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
               loss_pos_grad=pos2.grad.detach().clone())
    return out


def main():
    mm = R.reference_model_module()
    warnings.simplefilter("ignore")
    out = {}
    for name in FIXTURES:
        g = torch.load(os.path.join(OUT, name))
        r = second_order(mm, g)
        # the fixture's own fp64 energies / forces: the model rebuilt here is the one the fixture was made from
        assert (r["F"] - g["F64"]).abs().max().item() < 1e-10 * max(1.0, g["F64"].abs().max().item()), name
        out[name] = r
        worst = max(r["grads"].items(), key=lambda kv: kv[1].abs().max().item())
        print(name, "s", r["s"].item(), "|Hv|", r["Hv"].abs().max().item(), "params", len(r["grads"]), "largest", worst[0],
              worst[1].abs().max().item())
    torch.save(out, os.path.join(OUT, "second_order_ref.pt"))
    print("wrote", os.path.join(OUT, "second_order_ref.pt"), os.path.getsize(os.path.join(OUT, "second_order_ref.pt")), "bytes")


if __name__ == "__main__":
    main()
