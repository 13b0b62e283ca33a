"""The analytic second-order pass of the engine, checked without a GPU: the per-element bodies the HIP kernels run
(csrc/tn_hvp_math.h, compiled host-only) in the engine's schedule (tests/hvp_host_mirror.py) against the specification
oracle/tensornet_second_order.py in fp64 (itself pinned to autograd-of-autograd, tests/test_oracle.py)."""
import os
import shutil

import pytest
import torch

from oracle import tensornet_second_order as S2
from oracle import tensornet_torch as T

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


def _check(sd, hp, z, pos, batch, v, q=None, box=None, tol=2e-5, ge=None):
    from tests import hvp_host_mirror as HM

    sd64 = T.cast_state_dict(sd, torch.float64)
    ref = S2.force_term(sd64, hp, z, pos.double(), batch, v.double(), q=None if q is None else q.double(),
                        box=None if box is None else box.double(), ge=None if ge is None else ge.double())
    out = HM.force_term_mirror(sd, hp, z, pos, batch, v, q=q, box=box, ge=ge)
    if ge is None:  # (seeded: the mirror's s is the head's part only)
        assert abs(out["s"].item() - ref["s"].item()) < tol * max(1.0, abs(ref["s"].item()))
    # H v in the positions, and the forces out of the same geometry kernels (their value half)
    assert (out["F"].double() - ref["F"]).abs().max().item() < tol * ref["F"].abs().max().item()
    assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < 5 * tol * ref["Hv"].abs().max().item()
    assert set(out["ent"]) == set(ref["ent"])
    for k, r in ref["ent"].items():
        o = out["ent"][k].double().reshape(r.shape)
        assert torch.isfinite(o).all(), k
        scale = max(r.abs().max().item(), 1e-6)
        assert (o - r).abs().max().item() < tol * scale, (k, (o - r).abs().max().item(), scale)


@pytest.mark.parametrize("extra,use_q", [({}, True), (dict(equivariance_invariance_group="SO(3)"), False), (dict(cutoff_lower=1.2), True)])
def test_kernel_bodies_in_engine_schedule_match_specification(golden_dir, extra, use_q):
    tiny = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    hp = dict(T.hparams_from_args(tiny["args"]), **extra)
    v = torch.randn(tiny["pos"].shape, generator=torch.Generator().manual_seed(3))
    _check(tiny["state_dict"], hp, tiny["z"], tiny["pos"], tiny["batch"], v, q=tiny["q"] if use_q else None)
    # the same with the energy seed of one-pass training
    ge = torch.randn(int(tiny["batch"].max()) + 1, generator=torch.Generator().manual_seed(8))
    _check(tiny["state_dict"], hp, tiny["z"], tiny["pos"], tiny["batch"], v, q=tiny["q"] if use_q else None, ge=ge)


def test_kernel_bodies_periodic_box(golden_dir):
    tiny = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    f = torch.load(os.path.join(golden_dir, "tiny_pbc_ref.pt"))
    hp = T.hparams_from_args(tiny["args"])
    v = torch.randn(f["pos"].shape, generator=torch.Generator().manual_seed(5))
    _check(tiny["state_dict"], hp, f["z"], f["pos"], f["batch"], v, box=f["box"])
    _check(tiny["state_dict"], hp, f["z"], f["pos"], f["batch"], v, box=f["box"], ge=torch.tensor([0.8]))


@pytest.mark.parametrize("name,extra,sizes,charges", [
    ("so3-ragged-single-atom", dict(equivariance_invariance_group="SO(3)"), [9, 14, 1], False),
    ("one-layer-lower-cutoff", dict(num_layers=1, cutoff_lower=0.8, cutoff_upper=4.5), [13, 6], False),
    ("odd-widths-three-layers", dict(embedding_dimension=24, num_rbf=10, max_z=100, num_layers=3), [12, 2, 7], True),
])
def test_kernel_bodies_on_the_gpu_test_configurations(name, extra, sizes, charges):
    """the configurations of tests/test_gpu_hvp.py (scaled down), random-init models: ragged molecules incl. a single atom (a row
    with its self edge only), SO(3), one layer with a lower cutoff, three layers with widths that are no power of two."""
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, **extra)
    torch.manual_seed(11)
    model = create_model(dict(args))
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(700 + m, n_atoms=n)
        zs.append(torch.from_numpy(zz))
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    z, pos, batch = torch.cat(zs), torch.cat(ps), torch.cat(bs)
    q = torch.tensor([float(m % 3 - 1) for m in range(len(sizes))]) if charges else None
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    sd = {k: t.detach() for k, t in model.state_dict().items()}
    _check(sd, T.hparams_from_args(args), z, pos.float(), batch, v, q=q)


@pytest.mark.parametrize("fixture", ["et_tiny_ref.pt", "et_tiny_vc_ref.pt"])
def test_et_kernel_bodies_in_packed_layout_match_specification(golden_dir, fixture):
    """Equivariant Transformer: the bodies of csrc/tn_et_hvp_math.h in the engine's packed layouts and launch order
    (tests/et_hvp_host_mirror.py) against oracle/et_second_order.py in fp64: every parameter (mapped back the way _et_grads does),
    the forces out of the same geometry kernels, and H v."""
    from oracle import et_second_order as E2
    from oracle import et_torch as ET
    from tests import et_hvp_host_mirror as EM

    g = torch.load(os.path.join(golden_dir, fixture))
    hp = ET.hparams_from_args(g["args"])
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch = g["z"], g["pos"], g["batch"]
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    ref = E2.force_term(sd64, hp, z, pos.double(), batch, v.double())
    out = EM.force_term_mirror(g["state_dict"], hp, z, pos, batch, v)
    tol = 5e-5
    assert abs(out["s"].item() - ref["s"].item()) < tol * max(1.0, abs(ref["s"].item()))
    assert (out["F"].double() - ref["F"]).abs().max().item() < tol * ref["F"].abs().max().item()
    assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < tol * ref["Hv"].abs().max().item()
    mine = EM.state_dict_grads(out["ent"], g["state_dict"], hp)
    assert set(ref["grads"]) <= set(mine)
    for k, r in ref["grads"].items():
        o = mine[k].double().reshape(r.shape)
        assert torch.isfinite(o).all(), k
        assert (o - r).abs().max().item() < tol * max(r.abs().max().item(), 1e-6), k


@pytest.mark.parametrize("name,extra,sizes", [
    ("values-only-no-neighbour-embedding", dict(distance_influence="values", neighbor_embedding=False), [9, 14, 1]),
    ("no-filters", dict(distance_influence="none"), [12, 5]),
    ("keys-vector-cutoff-three-layers", dict(distance_influence="keys", vector_cutoff=True, num_layers=3), [7, 11, 2]),
])
def test_et_kernel_bodies_on_other_configurations(name, extra, sizes):
    """branches the two fixtures do not reach: no neighbour embedding, value filter only, no distance filter at all, ragged molecules
    incl. a single atom; random-init models, specification vs host run of the bodies in the packed layout."""
    from oracle import et_second_order as E2
    from oracle import et_torch as ET
    from tests import et_hvp_host_mirror as EM
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    args = dict(W.ET_TINY_ARGS, **extra)
    torch.manual_seed(23)
    model = create_model(dict(args))
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(900 + m, n_atoms=n)
        zs.append(torch.from_numpy(zz))
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    z, pos, batch = torch.cat(zs), torch.cat(ps).float(), torch.cat(bs)
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    sd = {k: t.detach() for k, t in model.state_dict().items()}
    hp = ET.hparams_from_args(args)
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in sd.items()}
    ref = E2.force_term(sd64, hp, z, pos.double(), batch, v.double())
    out = EM.force_term_mirror(sd, hp, z, pos, batch, v)
    tol = 5e-4  # fp32 against fp64 on random-init models whose head activations reach 1e3 (measured 2e-4 there, 1e-5 elsewhere)
    assert abs(out["s"].item() - ref["s"].item()) < tol * max(1.0, abs(ref["s"].item()))
    assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < tol * max(ref["Hv"].abs().max().item(), 1e-6)
    mine = EM.state_dict_grads(out["ent"], sd, hp)
    for k, r in ref["grads"].items():
        o = mine[k].double().reshape(r.shape)
        assert torch.isfinite(o).all(), k
        assert (o - r).abs().max().item() < tol * max(r.abs().max().item(), 1e-6), k


@pytest.mark.parametrize("fixture", ["tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_tn2_kernel_bodies_in_planned_schedule_match_specification(golden_dir, fixture):
    """TensorNet2 + Coulomb head: the bodies of csrc/tn_tn2_hvp_math.h (+ TensorNet's) in the planned launch order
    (tests/tn2_hvp_host_mirror.py) against oracle/tn2_second_order.py in fp64: every parameter, forces, H v; total charges with the
    all-to-all Coulomb sum, and the reaction-field branch in a periodic box."""
    from oracle import tn2_second_order as N2
    from oracle import tn2_torch as T2
    from tests import tn2_hvp_host_mirror as M2

    g = torch.load(os.path.join(golden_dir, fixture))
    hp = T2.hparams_from_args(g["args"])
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch = g["z"], g["pos"], g["batch"]
    q = g["q"] if g.get("q") is not None else None
    box = g["box"] if g.get("box") is not None else None
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    ref = N2.force_term(sd64, hp, z, pos.double(), batch, v.double(), box=None if box is None else box.double(),
                        q=None if q is None else q.double())
    out = M2.force_term_mirror(g["state_dict"], hp, z, pos, batch, v, box=box, q=q)
    tol = 5e-5
    assert abs(out["s"].item() - ref["s"].item()) < tol * max(1.0, abs(ref["s"].item()))
    assert (out["F"].double() - ref["F"]).abs().max().item() < tol * ref["F"].abs().max().item()
    assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < tol * ref["Hv"].abs().max().item()
    refg = N2.state_dict_grads(ref, sd64, hp)
    mine = N2.state_dict_grads(dict(ent={k: t.double() for k, t in out["ent"].items()}, extra={k: t.double() for k, t in out["extra"].items()}),
                               sd64, hp)
    assert set(refg) == set(mine)
    for k, r in refg.items():
        o = mine[k].reshape(r.shape)
        assert torch.isfinite(o).all(), k
        assert (o - r).abs().max().item() < tol * max(r.abs().max().item(), 1e-6), k


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_kernel_bodies_with_energy_seed_match_specification(golden_dir, fixture):
    """One-pass training: the bodies with the energy seed ge (head_dual's extra term, the head's last bias, the Coulomb head's
    adjoints) in the engine's schedule, against the specifications' gradient of  S = s - sum_m ge_m E_m  in fp64: every parameter
    and the position gradient."""
    g = torch.load(os.path.join(golden_dir, fixture))
    sd = g["state_dict"]
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in sd.items()}
    z, pos, batch = g["z"], g["pos"], g["batch"]
    q = g["q"] if g.get("q") is not None else None
    box = g["box"] if g.get("box") is not None else None
    nmol = int(batch.max()) + 1
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    ge = torch.randn(nmol, generator=torch.Generator().manual_seed(4))
    d = lambda t: None if t is None else t.double()
    if fixture.startswith("et_"):
        from oracle import et_second_order as E2
        from oracle import et_torch as ET
        from tests import et_hvp_host_mirror as EM

        hp = ET.hparams_from_args(g["args"])
        ref = E2.force_term(sd64, hp, z, pos.double(), batch, v.double(), ge=ge.double())
        out = EM.force_term_mirror(sd, hp, z, pos, batch, v, ge=ge)
        refg, mine = ref["grads"], EM.state_dict_grads(out["ent"], sd, hp)
    elif fixture.startswith("tn2_"):
        from oracle import tn2_second_order as N2
        from oracle import tn2_torch as T2
        from tests import tn2_hvp_host_mirror as M2

        hp = T2.hparams_from_args(g["args"])
        ref = N2.force_term(sd64, hp, z, pos.double(), batch, v.double(), box=d(box), q=d(q), ge=ge.double())
        out = M2.force_term_mirror(sd, hp, z, pos, batch, v, box=box, q=q, ge=ge)
        refg = N2.state_dict_grads(ref, sd64, hp)
        mine = N2.state_dict_grads(dict(ent={k: t.double() for k, t in out["ent"].items()}, extra={k: t.double() for k, t in out["extra"].items()}),
                                   sd64, hp)
    else:
        from oracle import tensornet_second_order as S2
        from oracle import tensornet_torch as T
        from tests import hvp_host_mirror as HM

        hp = T.hparams_from_args(g["args"])
        ref = S2.force_term(sd64, hp, z, pos.double(), batch, v.double(), q=d(q), ge=ge.double())
        out = HM.force_term_mirror(sd, hp, z, pos, batch, v, q=q, ge=ge)
        refg = S2.state_dict_grads(ref["ent"], sd64, hp)
        mine = S2.state_dict_grads({k: t.double() for k, t in out["ent"].items()}, sd64, hp)
    tol = 5e-5
    assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < tol * ref["Hv"].abs().max().item()
    checked = 0
    for k, r in refg.items():
        if r.abs().max() == 0:
            continue
        o = mine[k].double().reshape(r.shape)
        assert torch.isfinite(o).all(), k
        assert (o - r).abs().max().item() < tol * r.abs().max().item(), (k, (o - r).abs().max().item(), r.abs().max().item())
        checked += 1
    assert checked >= 30
@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_entry_to_parameter_mapping_against_the_reference_double_backward(golden_dir, fixture):
    """The product's mapping of engine gradient entries to parameters (models/model.py _tensornet_grads / _et_grads) fed with the
    host run of the kernel bodies (same entry names as the engine's layout), against the UNMODIFIED reference's own second autograd
    pass (tests/golden/second_order_ref.pt): every parameter the reference has a non-zero gradient for must come out, at fp32
    accuracy - what tests/test_gpu_hvp.py::test_engine_second_order_pass_equals_the_reference_double_backward checks with the
    engine's numbers on the GPU."""
    from torchmdnet_amd.models.model import create_model

    ref = torch.load(os.path.join(golden_dir, "second_order_ref.pt"))[fixture]
    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    z, pos, batch, v = g["z"], g["pos"], g["batch"], ref["v"].float()
    if fixture.startswith("et_"):
        from oracle import et_torch as ET
        from tests import et_hvp_host_mirror as EM

        out = EM.force_term_mirror(g["state_dict"], ET.hparams_from_args(g["args"]), z, pos, batch, v)
        grads = model._et_grads({k: t.reshape(-1) for k, t in out["ent"].items()})
    elif fixture.startswith("tn2_"):
        from oracle import tn2_torch as T2
        from tests import tn2_hvp_host_mirror as M2

        out = M2.force_term_mirror(g["state_dict"], T2.hparams_from_args(g["args"]), z, pos, batch, v, box=g.get("box"), q=g.get("q"))
        ent = {k: t.reshape(-1) for k, t in out["ent"].items()}
        R_ = "representation_model."
        F_, K_, qd_ = g["args"]["embedding_dimension"], g["args"]["num_rbf"], g["args"]["q_dim"]
        for l in range(g["args"]["num_layers"]):  # the engine's layout: the first edge layer as its three column blocks
            m0 = out["ent"][f"l{l}.M0"].reshape(F_, K_ + 2 * qd_)
            ent[f"l{l}.M0"], ent[f"l{l}.M0b"], ent[f"l{l}.M0c"] = (m0[:, :K_].reshape(-1), m0[:, K_:K_ + qd_].reshape(-1),
                                                                   m0[:, K_ + qd_:].reshape(-1))
        heads = [R_ + "charge_predict_0."] + [R_ + f"charge_predicts.{l}." for l in range(g["args"]["num_layers"])]
        for h, pre in enumerate(heads):  # the charge heads' entries by the engine's names
            for key, sk in (("ln_w", "q_norm.weight"), ("ln_b", "q_norm.bias"), ("W1", "q_mlp.layers.0.weight"), ("b1", "q_mlp.layers.0.bias"),
                            ("W2", "q_mlp.layers.2.weight"), ("b2", "q_mlp.layers.2.bias"), ("W3", "q_mlp.layers.4.weight"),
                            ("b3", "q_mlp.layers.4.bias")):
                ent[f"cp{h}.{key}"] = out["extra"][pre + sk].reshape(-1)
        grads = model._tn2_grads(ent)
    else:
        from oracle import tensornet_torch as T
        from tests import hvp_host_mirror as HM

        out = HM.force_term_mirror(g["state_dict"], T.hparams_from_args(g["args"]), z, pos, batch, v, q=g["q"])
        grads = model._tensornet_grads({k: t.reshape(-1) for k, t in out["ent"].items()})
    by_name = {id(p): k for k, p in model.named_parameters()}
    mine = {by_name[id(p)]: t.double() for p, t in grads.items()}
    tol, checked = 5e-5, 0
    for k, r in ref["grads"].items():
        if r.abs().max() == 0:
            continue
        assert k in mine, k
        assert (mine[k].reshape(r.shape) - r).abs().max().item() < tol * r.abs().max().item(), k
        checked += 1
    assert checked >= 30
    assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < tol * ref["Hv"].abs().max().item()


@pytest.mark.parametrize("name,extra,sizes,charges", [
    ("so3-one-layer-no-total-charge", dict(equivariance_invariance_group="SO(3)", num_layers=1, q_weights=[1.0, 0.5]), [9, 14], False),
    ("three-layers-small-q", dict(num_layers=3, q_dim=4, q_weights=[1.0, 0.5, 2.0, 1.5]), [7, 12, 1, 5], True),
    ("reaction-field-no-box", dict(coulomb_cutoff=4.0), [11, 8], True),
])
def test_tn2_kernel_bodies_on_other_configurations(name, extra, sizes, charges):
    """TensorNet2 branches the two fixtures do not reach: SO(3), one and three layers (q_weights per charge set), no total charge,
    a single-atom molecule, the reaction-field Coulomb term without a box; random-init models, specification vs host run of the bodies."""
    from oracle import tn2_second_order as N2
    from oracle import tn2_torch as T2
    from tests import tn2_hvp_host_mirror as M2
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=8, q_weights=[1.0, 0.5, 2.0], derivative=True)
    args.update(extra)
    torch.manual_seed(29)
    model = create_model(dict(args))
    with torch.no_grad():  # default init leaves the charge channels tiny: give every block weight
        for k, p in model.named_parameters():
            if "charge_predict" in k and p.dim() > 1:
                p.mul_(3.0)
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(1700 + m, n_atoms=n)
        zs.append(torch.from_numpy(zz))
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    z, pos, batch = torch.cat(zs), torch.cat(ps).float(), torch.cat(bs)
    q = torch.tensor([float(m % 3 - 1) for m in range(len(sizes))]) if charges else None
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    ge = torch.randn(len(sizes), generator=torch.Generator().manual_seed(6))
    sd = {k: t.detach() for k, t in model.state_dict().items()}
    hp = T2.hparams_from_args(args)
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in sd.items()}
    for seed in (None, ge):
        ref = N2.force_term(sd64, hp, z, pos.double(), batch, v.double(), q=None if q is None else q.double(),
                            ge=None if seed is None else seed.double())
        out = M2.force_term_mirror(sd, hp, z, pos, batch, v, q=q, ge=seed)
        tol = 2e-4
        assert (out["Hv"].double() - ref["Hv"]).abs().max().item() < tol * ref["Hv"].abs().max().item()
        refg = N2.state_dict_grads(ref, sd64, hp)
        mine = N2.state_dict_grads(dict(ent={k: t.double() for k, t in out["ent"].items()}, extra={k: t.double() for k, t in out["extra"].items()}),
                                   sd64, hp)
        for k, r in refg.items():
            if r.abs().max() == 0:
                continue
            o = mine[k].reshape(r.shape)
            assert torch.isfinite(o).all(), k
            assert (o - r).abs().max().item() < tol * r.abs().max().item(), (name, k, (o - r).abs().max().item(), r.abs().max().item())
