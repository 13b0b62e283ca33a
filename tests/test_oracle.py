"""not-gpu: pin the oracle.  Three independent restatements of the reference's TensorNet E+F path
   (a) oracle/tensornet_torch.py   full-tensor PyTorch, autograd forces
   (b) oracle/tensornet_adjoint.py kernel-level layout, HAND-DERIVED reverse pass (torch, vectorised)
   (c) oracle/tensornet_c.c        scalar C transliteration of (b)
are checked against the reference's own golden vector (tests/expected.pkl, committed as
tests/golden/expected_tensornet_scalar.pt) and against fixtures produced by the UNMODIFIED reference
(oracle/make_golden.py).  When /root/reference is present the live reference is compared as well."""
import os

import pytest
import torch

from oracle import ref_shims as R
from oracle import tensornet_adjoint as A
from oracle import tensornet_c as CO
from oracle import tensornet_torch as T


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.fixture(scope="module")
def tiny(golden_dir):
    return torch.load(os.path.join(golden_dir, "tiny_ref.pt"))


def _seeded_qm9_model(g):
    """Rebuild the golden test's model from the seed (bit-identical init order, see test_host.py)."""
    from torchmdnet_amd.models.model import create_model

    R.seed_everything(1234)
    model = create_model(dict(g["args"]))
    z, pos, batch = R.create_example_batch(n_atoms=5)
    return model, z, pos, batch


def test_reference_golden_vector_all_oracles(golden_dir):
    """tests/expected.pkl['tensornet']['Scalar'] at the reference's own tolerance (atol=rtol=1e-5)."""
    g = torch.load(os.path.join(golden_dir, "expected_tensornet_scalar.pt"))
    model, z, pos, batch = _seeded_qm9_model(g)
    assert torch.equal(pos, g["pos"])
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    hp = T.hparams_from_args(g["args"])
    for name, fn in [("torch", T.energy_and_forces), ("adjoint", A.energy_forces), ("c", CO.energy_forces)]:
        E, F = fn(sd, hp, z, pos, batch)
        torch.testing.assert_close(E, g["pred"], atol=1e-5, rtol=1e-5, msg=name)
        torch.testing.assert_close(F, g["deriv"], atol=1e-5, rtol=1e-5, msg=name)


@pytest.mark.parametrize("impl", ["torch", "adjoint", "c"])
def test_fp64_fixture(tiny, impl):
    """fp64: the restatements agree with the reference to round-off (1e-12 relative)."""
    fn = {"torch": T.energy_and_forces, "adjoint": A.energy_forces, "c": CO.energy_forces}[impl]
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    hp = T.hparams_from_args(tiny["args"])
    E, F = fn(sd64, hp, tiny["z"], tiny["pos"].double(), tiny["batch"], q=tiny["q"].double())
    assert rel_err(E, tiny["E64"]) < 1e-12
    assert rel_err(F, tiny["F64"]) < 1e-12


@pytest.mark.parametrize("impl", ["torch", "adjoint", "c"])
def test_fp32_fixture_and_charges(tiny, impl):
    fn = {"torch": T.energy_and_forces, "adjoint": A.energy_forces, "c": CO.energy_forces}[impl]
    hp = T.hparams_from_args(tiny["args"])
    E, F = fn(tiny["state_dict"], hp, tiny["z"], tiny["pos"], tiny["batch"], q=tiny["q"])
    assert rel_err(E, tiny["E"]) < 1e-5 and rel_err(F, tiny["F"]) < 1e-5
    E, F = fn(tiny["state_dict"], hp, tiny["z"], tiny["pos"], tiny["batch"])
    assert rel_err(E, tiny["E_q0"]) < 1e-5 and rel_err(F, tiny["F_q0"]) < 1e-5


def test_intermediates(tiny):
    hp = T.hparams_from_args(tiny["args"])
    x, inter = T.tensornet_representation(tiny["state_dict"], hp, tiny["z"], tiny["pos"], tiny["batch"], q=tiny["q"],
                                          return_intermediates=True)
    for k in ["X_embed", "X_layer0", "X_layer1"]:
        assert rel_err(inter[k], tiny["inter"][k]) < 1e-5
    assert rel_err(x, tiny["inter"]["x"]) < 1e-5
    _, _, c = A.energy_forces(tiny["state_dict"], hp, tiny["z"], tiny["pos"], tiny["batch"], q=tiny["q"], want_cache=True)
    assert rel_err(A.compose(c["X_embed"]), tiny["inter"]["X_embed"]) < 1e-5
    assert rel_err(A.compose(c["X_final"]), tiny["inter"]["X_layer1"]) < 1e-5


@pytest.mark.parametrize("impl", ["torch", "adjoint", "c"])
def test_triclinic_periodic_fixture(tiny, golden_dir, impl):
    fn = {"torch": T.energy_and_forces, "adjoint": A.energy_forces, "c": CO.energy_forces}[impl]
    p = torch.load(os.path.join(golden_dir, "tiny_pbc_ref.pt"))
    hp = T.hparams_from_args(tiny["args"])
    E, F = fn(tiny["state_dict"], hp, p["z"], p["pos"], p["batch"], box=p["box"])
    assert rel_err(E, p["E"]) < 1e-5 and rel_err(F, p["F"]) < 1e-5


def test_c2_fixture(golden_dir):
    """BASELINE configs[1] weights (seed 0), 4 molecules of S-mol64, vs the unmodified reference."""
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "c2_ref.pt"))
    torch.manual_seed(0)
    model = create_model(dict(W.C2_ARGS))
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    z, pos, batch = W.synthetic_batch(n_mol=1)
    hp = T.hparams_from_args(W.C2_ARGS)
    E, F = CO.energy_forces(sd, hp, z, pos, batch)
    assert rel_err(E, g["E"][:1]) < 1e-5 and rel_err(F, g["F"][:64]) < 1e-5
    E, F = T.energy_and_forces(sd, hp, z, pos, batch)
    assert rel_err(E, g["E"][:1]) < 1e-5 and rel_err(F, g["F"][:64]) < 1e-5


def test_hand_adjoint_equals_autograd_so3_and_lower_cutoff(tiny):
    """configurations the fixtures do not cover: SO(3) group and cutoff_lower > 0 (fp64, round-off)."""
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    for extra in [dict(equivariance_invariance_group="SO(3)"), dict(cutoff_lower=1.2)]:
        hp = dict(T.hparams_from_args(tiny["args"]), **extra)
        Ea, Fa = T.energy_and_forces(sd64, hp, tiny["z"], tiny["pos"].double(), tiny["batch"], q=tiny["q"].double())
        for fn in (A.energy_forces, CO.energy_forces):
            E, F = fn(sd64, hp, tiny["z"], tiny["pos"].double(), tiny["batch"], q=tiny["q"].double())
            assert rel_err(E, Ea) < 1e-11 and rel_err(F, Fa) < 1e-11


def test_forces_are_minus_energy_gradient_finite_difference(tiny):
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    hp = T.hparams_from_args(tiny["args"])
    z, pos, batch = tiny["z"][:7], tiny["pos"][:7].double(), tiny["batch"][:7]
    _, F = CO.energy_forces(sd64, hp, z, pos, batch)
    h = 1e-5
    for (i, a) in [(0, 0), (3, 2), (6, 1)]:
        pp, pm = pos.clone(), pos.clone()
        pp[i, a] += h
        pm[i, a] -= h
        Ep, _ = CO.energy_forces(sd64, hp, z, pp, batch, want_forces=False)
        Em, _ = CO.energy_forces(sd64, hp, z, pm, batch, want_forces=False)
        fd = -(Ep.sum() - Em.sum()) / (2 * h)
        assert abs(fd.item() - F[i, a].item()) < 1e-7 * max(1.0, abs(fd.item()))


@pytest.mark.skipif(not R.reference_available(), reason="/root/reference not present (GPU box)")
def test_live_reference_random_system():
    """The unmodified reference, imported here, on a fresh random system (not a committed fixture)."""
    from torchmdnet_amd import workloads as W

    mm = R.reference_model_module()
    torch.manual_seed(11)
    args = dict(W.TINY_ARGS, num_layers=3)
    ref = mm.create_model(dict(args))
    z, pos, batch = W.synthetic_batch(n_mol=2, n_atoms=17, first_seed=900)
    q = torch.tensor([0.5, -2.0])
    Er, Fr = ref(z, pos.clone(), batch, q=q)
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    hp = T.hparams_from_args(args)
    for fn in (T.energy_and_forces, A.energy_forces, CO.energy_forces):
        E, F = fn(sd, hp, z, pos, batch, q=q)
        assert rel_err(E, Er.detach()) < 1e-5 and rel_err(F, Fr.detach()) < 1e-5


# ----------------------------------------------------------------------------------------------
# Equivariant Transformer: oracle/et_torch.py against the unmodified reference
# ----------------------------------------------------------------------------------------------
from oracle import et_torch as ET  # noqa: E402


@pytest.mark.parametrize("fixture", ["et_tiny_ref.pt", "et_tiny_vc_ref.pt"])
def test_et_oracle_fixtures(golden_dir, fixture):
    """fp64: round-off agreement with the reference (both branch sets: cutoff on attention vs on values, keys-only
    distance influence); fp32: the reference's tolerance; per-layer dx / dvec through the residual stream."""
    g = torch.load(os.path.join(golden_dir, fixture))
    hp = ET.hparams_from_args(g["args"])
    sd64 = T.cast_state_dict(g["state_dict"], torch.float64)
    E, F = ET.energy_and_forces(sd64, hp, g["z"], g["pos"].double(), g["batch"])
    assert rel_err(E, g["E64"]) < 1e-12 and rel_err(F, g["F64"]) < 1e-12
    E, F = ET.energy_and_forces(g["state_dict"], hp, g["z"], g["pos"], g["batch"])
    assert rel_err(E, g["E"]) < 1e-5 and rel_err(F, g["F"]) < 1e-5
    x, vec, inter = ET.et_representation(g["state_dict"], hp, g["z"], g["pos"], g["batch"], return_intermediates=True)
    assert rel_err(x, g["inter"]["x_out"]) < 1e-5 and rel_err(vec, g["inter"]["vec_out"]) < 1e-5
    xprev, vprev = inter["x_embed"], torch.zeros_like(vec)
    for l in range(hp["num_layers"]):
        assert rel_err(inter[f"x_layer{l}"] - xprev, g["inter"][f"dx_layer{l}"]) < 1e-4
        assert rel_err(inter[f"vec_layer{l}"] - vprev, g["inter"][f"dvec_layer{l}"]) < 1e-4
        xprev, vprev = inter[f"x_layer{l}"], inter[f"vec_layer{l}"]


@pytest.mark.parametrize("fixture", ["et_tiny_ref.pt", "et_tiny_vc_ref.pt"])
def test_et_hand_derived_reverse_pass(golden_dir, fixture):
    """oracle/et_adjoint.py (kernel-level spec, explicit adjoints + forward tangents) == reference in fp64."""
    from oracle import et_adjoint as EA

    g = torch.load(os.path.join(golden_dir, fixture))
    hp = ET.hparams_from_args(g["args"])
    sd64 = T.cast_state_dict(g["state_dict"], torch.float64)
    E, F = EA.energy_forces(sd64, hp, g["z"], g["pos"].double(), g["batch"])
    assert rel_err(E, g["E64"]) < 1e-12 and rel_err(F, g["F64"]) < 1e-11


def test_et_reference_golden_vector(golden_dir):
    """tests/expected.pkl['equivariant-transformer']['Scalar'] at the reference's own tolerance (atol=rtol=1e-5);
    weights regenerated from the seed through the host module (bit-identical init order, tests/test_host.py)."""
    from oracle import et_adjoint as EA
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "expected_et_scalar.pt"))
    R.seed_everything(1234)
    model = create_model(dict(g["args"]))
    z, pos, batch = R.create_example_batch(n_atoms=5)
    assert torch.equal(pos, g["pos"])
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    hp = ET.hparams_from_args(g["args"])
    for name, fn in [("torch", ET.energy_and_forces), ("adjoint", EA.energy_forces)]:
        E, F = fn(sd, hp, z, pos, batch)
        torch.testing.assert_close(E, g["pred"], atol=1e-5, rtol=1e-5, msg=name)
        torch.testing.assert_close(F, g["deriv"], atol=1e-5, rtol=1e-5, msg=name)


@pytest.mark.parametrize("fixture", ["tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_tensornet2_oracle_matches_reference_fixture(golden_dir, fixture):
    """oracle/tn2_torch.py (TensorNet2 + ScalarPlusWeightedCoulomb restatement) against fixtures written by the UNMODIFIED
    reference (oracle/make_golden_tn2.py): energies, forces, every layer's tensor features and charge channels, fp64 1e-10;
    all-to-all Coulomb with total charges on ragged molecules, and the reaction-field branch in a triclinic periodic box."""
    from oracle import tn2_torch as T2
    from oracle import tensornet_torch as T

    g = torch.load(os.path.join(golden_dir, fixture))
    hp = T2.hparams_from_args(g["args"])
    sd64 = T.cast_state_dict(g["state_dict"], torch.float64)
    box = None if g["box"] is None else g["box"].double()
    inter = {}
    pos = g["pos"].double().requires_grad_(True)
    y = T2.energy(sd64, hp, g["z"], pos, g["batch"], box=box, q=g["q"].double(), inter=inter)
    (dy,) = torch.autograd.grad([y], [pos], grad_outputs=[torch.ones_like(y)])
    assert (y - g["E64"]).abs().max() < 1e-10 and (-dy - g["F64"]).abs().max() < 1e-10
    for k, v in g["inter64"].items():
        assert (inter[k] - v).abs().max() < 1e-10, k
    # fp32 evaluation agrees with the reference's fp32 run to rounding
    E, F = T2.energy_and_forces(g["state_dict"], hp, g["z"], g["pos"], g["batch"], box=g["box"], q=g["q"])
    assert (E - g["E"]).abs().max() < 2e-5 * g["E"].abs().max() and (F - g["F"]).abs().max() < 2e-5 * g["F"].abs().max()


# ---------------------------------------------------------------------------------------------- second-order pass (force matching)
from oracle import tensornet_second_order as S2  # noqa: E402


def _autograd_force_term(sd, hp, z, pos, batch, v, q=None, box=None):
    """d/d theta and d/d pos of  s = v . d(sum E)/d pos  by differentiating the autograd oracle twice (what the reference does:
    model.py:618-628 with create_graph=True)."""
    keys = [k for k, t in sd.items() if t.dtype == pos.dtype and t.dim() > 0 and "distance" not in k and "prior" not in k]
    sdg = {k: (t.clone().requires_grad_(True) if k in keys else t) for k, t in sd.items()}
    p = pos.clone().requires_grad_(True)
    E = T.energy(sdg, hp, z, p, batch, box=box, q=q)
    (gp,) = torch.autograd.grad(E.sum(), p, create_graph=True)
    s = (gp * v).sum()
    grads = torch.autograd.grad(s, [sdg[k] for k in keys] + [p], allow_unused=True)
    return s.detach(), {k: (torch.zeros_like(sd[k]) if g is None else g) for k, g in zip(keys, grads[:-1])}, grads[-1]


@pytest.mark.parametrize("extra,use_q", [({}, True), (dict(equivariance_invariance_group="SO(3)"), False), (dict(cutoff_lower=1.2), True)])
def test_hand_second_order_pass_equals_autograd_of_autograd(tiny, extra, use_q):
    """oracle/tensornet_second_order.py (the specification of the engine's analytic force-matching pass) against the reference's
    way of getting the same numbers - two nested autograd passes - in fp64: every parameter, and H v in the positions."""
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    hp = dict(T.hparams_from_args(tiny["args"]), **extra)
    z, pos, batch = tiny["z"], tiny["pos"].double(), tiny["batch"]
    q = tiny["q"].double() if use_q else None
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    s, ref, Hv = _autograd_force_term(sd64, hp, z, pos, batch, v, q=q)
    out = S2.force_term(sd64, hp, z, pos, batch, v, q=q)
    assert abs(out["s"].item() - s.item()) < 1e-12 * max(1.0, abs(s.item()))
    assert abs(out["s"].item() + (v * out["F"]).sum().item()) < 1e-12  # s = - v . F
    assert rel_err(out["Hv"], Hv) < 1e-11
    mine = S2.state_dict_grads(out["ent"], sd64, hp)
    assert set(ref) <= set(mine)
    for k, r in ref.items():
        if r.abs().max() > 0:
            assert rel_err(mine[k].reshape(r.shape), r) < 1e-10, k
        else:
            assert mine[k].abs().max() < 1e-12, k


def test_hand_second_order_pass_periodic(tiny, golden_dir):
    f = torch.load(os.path.join(golden_dir, "tiny_pbc_ref.pt"))
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    hp = T.hparams_from_args(tiny["args"])
    z, pos, batch, box = f["z"], f["pos"].double(), f["batch"], f["box"].double()
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    s, ref, Hv = _autograd_force_term(sd64, hp, z, pos, batch, v, box=box)
    out = S2.force_term(sd64, hp, z, pos, batch, v, box=box)
    assert abs(out["s"].item() - s.item()) < 1e-12 * max(1.0, abs(s.item())) and rel_err(out["Hv"], Hv) < 1e-11
    mine = S2.state_dict_grads(out["ent"], sd64, hp)
    for k, r in ref.items():
        if r.abs().max() > 0:
            assert rel_err(mine[k].reshape(r.shape), r) < 1e-10, k


@pytest.mark.parametrize("fixture", ["et_tiny_ref.pt", "et_tiny_vc_ref.pt"])
def test_et_hand_second_order_pass_equals_autograd_of_autograd(golden_dir, fixture):
    """oracle/et_second_order.py - the specification of an analytic force-matching pass for the Equivariant Transformer (not built in
    the engine yet: DESIGN.md 9b) - against two nested autograd passes over oracle/et_torch.py in fp64: every parameter and H v
    (both distance filters + neighbour embedding; vector cutoff + key filter only)."""
    from oracle import et_second_order as E2

    g = torch.load(os.path.join(golden_dir, fixture))
    hp = ET.hparams_from_args(g["args"])
    sd = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch = g["z"], g["pos"].double(), g["batch"]
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    keys = [k for k, t in sd.items() if t.is_floating_point() and t.dim() > 0 and "distance" not in k.split("neighbor_embedding.")[0]]
    sdg = {k: (t.clone().requires_grad_(True) if k in keys else t) for k, t in sd.items()}
    p = pos.clone().requires_grad_(True)
    (gp,) = torch.autograd.grad(ET.energy(sdg, hp, z, p, batch).sum(), p, create_graph=True)
    s = (gp * v).sum()
    grads = torch.autograd.grad(s, [sdg[k] for k in keys] + [p], allow_unused=True)
    out = E2.force_term(sd, hp, z, pos, batch, v)
    assert abs(out["s"].item() - s.item()) < 1e-11 * max(1.0, abs(s.item()))
    assert rel_err(out["F"], -gp.detach()) < 1e-11 and rel_err(out["Hv"], grads[-1]) < 1e-11
    checked = 0
    for k, r in zip(keys, grads[:-1]):
        if r is None or r.abs().max() == 0:
            assert k not in out["grads"] or out["grads"][k].abs().max() < 1e-12, k
            continue
        assert rel_err(out["grads"][k].reshape(r.shape), r) < 1e-10, k
        checked += 1
    assert checked >= 30


@pytest.mark.parametrize("fixture", ["tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_tn2_hand_second_order_pass_equals_autograd_of_autograd(golden_dir, fixture):
    """oracle/tn2_second_order.py - a hand-derived E + F program of TensorNet2 + ScalarPlusWeightedCoulomb with its tangent (the
    specification of an analytic force-matching pass; the engine keeps the difference quotient for this architecture) - against two
    nested autograd passes over oracle/tn2_torch.py in fp64: energies, forces, every parameter, H v; all-to-all Coulomb with total
    charges on ragged molecules, and the reaction-field branch in a periodic box."""
    from oracle import tn2_second_order as N2
    from oracle import tn2_torch as T2

    g = torch.load(os.path.join(golden_dir, fixture))
    hp = T2.hparams_from_args(g["args"])
    sd = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch = g["z"], g["pos"].double(), g["batch"]
    q = g["q"].double() if g.get("q") is not None else None
    box = g["box"].double() if g.get("box") is not None else None
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    keys = [k for k, t in sd.items() if t.is_floating_point() and t.dim() > 0 and "distance" not in k and "qweights" not in k and "prior" not in k]
    sdg = {k: (t.clone().requires_grad_(True) if k in keys else t) for k, t in sd.items()}
    p = pos.clone().requires_grad_(True)
    E = T2.energy(sdg, hp, z, p, batch, box=box, q=q)
    (gp,) = torch.autograd.grad(E.sum(), p, create_graph=True)
    s = (gp * v).sum()
    grads = torch.autograd.grad(s, [sdg[k] for k in keys] + [p], allow_unused=True)
    out = N2.force_term(sd, hp, z, pos, batch, v, box=box, q=q)
    assert rel_err(out["E"], E.detach()) < 1e-12 and rel_err(out["F"], -gp.detach()) < 1e-11
    assert abs(out["s"].item() - s.item()) < 1e-11 * max(1.0, abs(s.item())) and rel_err(out["Hv"], grads[-1]) < 1e-11
    mine = N2.state_dict_grads(out, sd, hp)
    checked = 0
    for k, r in zip(keys, grads[:-1]):
        if r is None or r.abs().max() == 0:
            assert k not in mine or mine[k].abs().max() < 1e-12, k
            continue
        assert rel_err(mine[k].reshape(r.shape), r) < 1e-10, k
        checked += 1
    assert checked >= 60


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_second_order_specifications_equal_the_reference_double_backward(golden_dir, fixture):
    """The three specifications of the analytic force-matching pass against the UNMODIFIED reference's own second autograd pass
    (model.py:618-628 in training mode, fp64; tests/golden/second_order_ref.pt from oracle/make_golden_second_order.py): s = - v . F,
    H v = d s / d pos and d s / d theta for every parameter the reference gave a gradient."""
    ref = torch.load(os.path.join(golden_dir, "second_order_ref.pt"))[fixture]
    g = torch.load(os.path.join(golden_dir, fixture))
    sd = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch, v = g["z"], g["pos"].double(), g["batch"], ref["v"]
    q = g["q"].double() if g.get("q") is not None else None
    box = g["box"].double() if g.get("box") is not None else None
    if fixture.startswith("et_"):
        from oracle import et_second_order as E2

        out = E2.force_term(sd, ET.hparams_from_args(g["args"]), z, pos, batch, v)
        mine = out["grads"]
    elif fixture.startswith("tn2_"):
        from oracle import tn2_second_order as N2
        from oracle import tn2_torch as T2

        hp = T2.hparams_from_args(g["args"])
        out = N2.force_term(sd, hp, z, pos, batch, v, box=box, q=q)
        mine = N2.state_dict_grads(out, sd, hp)
    else:
        hp = T.hparams_from_args(g["args"])
        out = S2.force_term(sd, hp, z, pos, batch, v, q=q)
        mine = S2.state_dict_grads(out["ent"], sd, hp)
    assert rel_err(out["F"], ref["F"]) < 1e-10
    assert abs(out["s"].item() - ref["s"].item()) < 1e-10 * max(1.0, abs(ref["s"].item()))
    assert rel_err(out["Hv"], ref["Hv"]) < 1e-10
    checked = 0
    for k, r in ref["grads"].items():
        if r.abs().max() == 0:
            assert k not in mine or mine[k].abs().max() < 1e-12, k
            continue
        assert k in mine, k
        assert rel_err(mine[k].reshape(r.shape), r) < 1e-9, k
        checked += 1
    assert checked >= 30


def test_hand_second_order_pass_with_energy_seed(tiny):
    """One-pass training: with ge = d loss / d E the specification returns the gradient of  S = v . d(sum E)/d pos - sum_m ge_m E_m
    (parameters and positions) - against autograd of that scalar in fp64."""
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    hp = T.hparams_from_args(tiny["args"])
    z, pos, batch, q = tiny["z"], tiny["pos"].double(), tiny["batch"], tiny["q"].double()
    nmol = int(batch.max()) + 1
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    ge = torch.randn(nmol, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    keys = [k for k, t in sd64.items() if t.dtype == pos.dtype and t.dim() > 0 and "distance" not in k and "prior" not in k]
    sdg = {k: (t.clone().requires_grad_(True) if k in keys else t) for k, t in sd64.items()}
    p = pos.clone().requires_grad_(True)
    E = T.energy(sdg, hp, z, p, batch, q=q)
    (gp,) = torch.autograd.grad(E.sum(), p, create_graph=True)
    S = (gp * v).sum() - (ge * E.view(-1)).sum()
    grads = torch.autograd.grad(S, [sdg[k] for k in keys] + [p], allow_unused=True)
    out = S2.force_term(sd64, hp, z, pos, batch, v, q=q, ge=ge)
    assert abs(out["s"].item() - S.item()) < 1e-12 * max(1.0, abs(S.item()))
    assert rel_err(out["Hv"], grads[-1]) < 1e-11
    mine = S2.state_dict_grads(out["ent"], sd64, hp)
    for k, r in zip(keys, grads[:-1]):
        if r is not None and r.abs().max() > 0:
            assert rel_err(mine[k].reshape(r.shape), r) < 1e-10, k


@pytest.mark.parametrize("fixture", ["et_tiny_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_et_tn2_second_order_pass_with_energy_seed(golden_dir, fixture):
    """One-pass training for the Equivariant Transformer and TensorNet2 (Coulomb head included): gradient of
    S = v . d(sum E)/d pos - sum_m ge_m E_m  in the parameters and the positions against autograd of that scalar in fp64."""
    g = torch.load(os.path.join(golden_dir, fixture))
    sd = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch = g["z"], g["pos"].double(), g["batch"]
    q = g["q"].double() if g.get("q") is not None else None
    box = g["box"].double() if g.get("box") is not None else None
    nmol = int(batch.max()) + 1
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    ge = torch.randn(nmol, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    et = fixture.startswith("et_")
    if et:
        from oracle import et_second_order as E2

        hp = ET.hparams_from_args(g["args"])
        keys = [k for k, t in sd.items() if t.is_floating_point() and t.dim() > 0 and "distance" not in k.split("neighbor_embedding.")[0]]
        energy = lambda sdg, p: ET.energy(sdg, hp, z, p, batch)
    else:
        from oracle import tn2_second_order as N2
        from oracle import tn2_torch as T2

        hp = T2.hparams_from_args(g["args"])
        keys = [k for k, t in sd.items() if t.is_floating_point() and t.dim() > 0 and "distance" not in k and "qweights" not in k and "prior" not in k]
        energy = lambda sdg, p: T2.energy(sdg, hp, z, p, batch, box=box, q=q)
    sdg = {k: (t.clone().requires_grad_(True) if k in keys else t) for k, t in sd.items()}
    p = pos.clone().requires_grad_(True)
    E = energy(sdg, p)
    (gp,) = torch.autograd.grad(E.sum(), p, create_graph=True)
    S = (gp * v).sum() - (ge * E.view(-1)).sum()
    grads = torch.autograd.grad(S, [sdg[k] for k in keys] + [p], allow_unused=True)
    if et:
        out = E2.force_term(sd, hp, z, pos, batch, v, ge=ge)
        mine = out["grads"]
    else:
        out = N2.force_term(sd, hp, z, pos, batch, v, box=box, q=q, ge=ge)
        mine = N2.state_dict_grads(out, sd, hp)
    assert abs(out["s"].item() - S.item()) < 1e-11 * max(1.0, abs(S.item()))
    assert rel_err(out["Hv"], grads[-1]) < 1e-11
    checked = 0
    for k, r in zip(keys, grads[:-1]):
        if r is None or r.abs().max() == 0:
            continue
        assert rel_err(mine[k].reshape(r.shape), r) < 1e-10, k
        checked += 1
    assert checked >= 30


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_seeded_specifications_equal_the_reference_backward_of_an_energy_and_force_loss(golden_dir, fixture):
    """loss = sum_m ge_m E_m + sum_i v_i . F_i back-propagated ONCE through the unmodified reference (training mode, fp64;
    tests/golden/second_order_ref.pt) against the seeded specifications: d loss / d theta = - d S / d theta and d loss / d pos =
    - d S / d pos with S = v . d(sum E)/d pos - sum_m ge_m E_m  - the identity one-pass training rests on (tmdnet_loss_param_grads)."""
    ref = torch.load(os.path.join(golden_dir, "second_order_ref.pt"))[fixture]
    g = torch.load(os.path.join(golden_dir, fixture))
    sd = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch, v, ge = g["z"], g["pos"].double(), g["batch"], ref["v"], ref["ge"]
    q = g["q"].double() if g.get("q") is not None else None
    box = g["box"].double() if g.get("box") is not None else None
    if fixture.startswith("et_"):
        from oracle import et_second_order as E2

        out = E2.force_term(sd, ET.hparams_from_args(g["args"]), z, pos, batch, v, ge=ge)
        mine = out["grads"]
    elif fixture.startswith("tn2_"):
        from oracle import tn2_second_order as N2
        from oracle import tn2_torch as T2

        hp = T2.hparams_from_args(g["args"])
        out = N2.force_term(sd, hp, z, pos, batch, v, box=box, q=q, ge=ge)
        mine = N2.state_dict_grads(out, sd, hp)
    else:
        hp = T.hparams_from_args(g["args"])
        out = S2.force_term(sd, hp, z, pos, batch, v, q=q, ge=ge)
        mine = S2.state_dict_grads(out["ent"], sd, hp)
    assert rel_err(-out["Hv"], ref["loss_pos_grad"]) < 1e-10
    checked = 0
    for k, r in ref["loss_grads"].items():
        if r.abs().max() == 0:
            continue
        assert k in mine, k
        assert rel_err(-mine[k].reshape(r.shape), r) < 1e-9, k
        checked += 1
    assert checked >= 30
