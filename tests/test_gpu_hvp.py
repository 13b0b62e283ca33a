"""-m gpu: the analytic second-order pass of force-matching training (tmdnet_force_param_grads, csrc/tn_hvp*.hip) through the C
ABI against its specification oracle/tensornet_second_order.py in fp64 (pinned to autograd-of-autograd, tests/test_oracle.py) - what
the reference computes with its second autograd pass (model.py:618-628, create_graph=True).  Bound: the north-star 1e-4 relative to
each tensor's largest entry.  Every case also walks the engine's intermediates (tmdnet_hvp_debug_tensor) against the host run of the
same kernel bodies (tests/hvp_host_mirror.py) and writes the table to gpurun_out/: the first mismatching name is the launch to
look at."""
import ctypes as C
import json
import os

import pytest
import torch

from torchmdnet_amd import workloads as W

pytestmark = pytest.mark.gpu
REL = 1e-4


def _ragged(sizes, seed):
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(seed + m, n_atoms=n)
        zs.append(torch.from_numpy(zz))
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    return torch.cat(zs), torch.cat(ps), torch.cat(bs)


def _walk_buffers(model, sd, hp, z, pos, batch, v, q):
    """engine intermediates vs the host run of the same bodies, in schedule order: [(name, rel err)]"""
    from tests import hvp_host_mirror as HM
    from torchmdnet_amd import _C

    mir = HM.force_term_mirror(sd, hp, z, pos, batch, v, q=q)
    L, st = _C.lib(), model._engine
    rows = []
    for name in mir["order"]:
        ref = mir["bufs"][name]
        n = L.tmdnet_hvp_debug_tensor(st.handle, None, name.encode(), None, 0)
        if n != ref.numel():
            rows.append((name, f"size {n} != {ref.numel()}"))
            continue
        out = torch.empty(ref.numel(), dtype=torch.float32, device="cuda")
        rc = L.tmdnet_hvp_debug_tensor(st.handle, None, name.encode(), C.c_void_p(out.data_ptr()), out.numel())
        assert rc == 0, L.tmdnet_last_error(st.handle).decode()
        torch.cuda.synchronize()
        o, r = out.cpu(), ref.reshape(-1)
        if name in ("gq", "gq_t"):  # the self-pair rows of the two direction blocks are never written (nor read)
            P1 = mir["P"] + 1
            keep = torch.ones(2, P1, ref.numel() // (2 * P1), dtype=torch.bool)
            keep[:, P1 - 1] = False
            o, r = o[keep.reshape(-1)], r[keep.reshape(-1)]
        err = (o - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        rows.append((name, err if torch.isfinite(o).all() else float("inf")))
    return rows


@pytest.mark.parametrize("name,extra,sizes,charges", [
    ("tiny-o3-charges", dict(), [18, 30, 4, 11, 1], True),
    ("so3", dict(equivariance_invariance_group="SO(3)"), [9, 21, 14], False),
    ("wide", dict(embedding_dimension=128, num_rbf=32, num_layers=2), [40, 33, 64], True),
    ("one-layer-lower-cutoff", dict(num_layers=1, cutoff_lower=0.8, cutoff_upper=4.5), [25, 12], False),
    ("odd-widths", dict(embedding_dimension=64, num_rbf=50, max_z=100, num_layers=3), [64, 64, 37, 64, 2], True),
])
def test_analytic_force_term_gradients_match_specification(hip_lib, name, extra, sizes, charges):
    from oracle import tensornet_second_order as S2
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, **extra)
    torch.manual_seed(11)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = _ragged(sizes, seed=700)
    B = len(sizes)
    q = torch.tensor([float(m % 3 - 1) for m in range(B)]) if charges else None
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), None, None if q is None else q.cuda(), B, v.cuda(),
                                                     want_hv=True)
    torch.cuda.synchronize()
    sd = {k: t.detach().cpu() for k, t in model.state_dict().items()}
    hp = T.hparams_from_args(args)
    rows = _walk_buffers(model, sd, hp, z, pos, batch, v, q)
    sd64 = T.cast_state_dict(sd, torch.float64)
    ref = S2.force_term(sd64, hp, z, pos.double(), batch, v.double(), q=None if q is None else q.double())
    refg = S2.state_dict_grads(ref["ent"], sd64, hp)
    by_name = {id(p): k for k, p in model.named_parameters()}
    errs = {}
    for p, g in grads.items():
        key = by_name[id(p)]
        r = refg[key].reshape(g.shape)
        scale = r.abs().max().item()
        errs[key] = (g.cpu().double() - r).abs().max().item() / scale if scale > 0 else g.abs().max().item()
    # s = - v . F: the same number from the engine's O2 gradient (d s / d O2 . O2 = s) and from the specification
    O2 = model.output_model.output_network.layers[2].weight
    s_engine = (grads[O2].cpu().double().reshape(-1) * O2.detach().cpu().double().reshape(-1)).sum().item()
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()  # H v = d s / d pos
    first_bad = next(((n, e) for n, e in rows if not (isinstance(e, float) and e < 1e-3)), None)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_{name}.json", "w") as fh:
        json.dump({"case": name, "s_engine": s_engine, "s_spec": ref["s"].item(), "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]),
                   "param_errors": errs, "first_bad_buffer": first_bad, "buffers": rows}, fh, indent=1)
    assert first_bad is None, first_bad
    assert abs(s_engine - ref["s"].item()) < REL * max(1.0, abs(ref["s"].item()))
    assert hv_err < REL, hv_err
    missing = {k for k, t in refg.items() if t.abs().max() > 0} - set(errs)
    assert not missing, missing
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert not bad, bad


@pytest.mark.parametrize("one_pass", [True, False])
def test_force_matching_backward_takes_the_analytic_pass(hip_lib, one_pass):
    """one_pass = True (the default): ONE second-order pass seeded with d loss / d E (tmdnet_loss_param_grads) delivers the energy
    term's gradient as well; False: first-order pass + second-order pass.  Same bound for both.
    derivative=True + parameter_gradients=True with force_gradient_order = 0: loss(E, F).backward() fills the weights' .grad
    with the energy term's exact gradient plus the analytic force term, and pos.grad with - g_E F - H g_F, against the double
    backward of the oracle in fp64 at the first-order pass's bound (the difference-quotient orders stay at 1e-3 / 2e-3 and
    leave the position term out, test_gpu_train.py)."""
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    def _oracle_force_matching_grads(model, args, z, pos, batch, R, ge):
        """d/d theta and d/d pos of  sum_i R_i . F_i + sum_m ge_m E_m  by double backward over the oracle in fp64"""
        sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
        p = pos.double().clone().requires_grad_(True)
        y = T.energy(sd, T.hparams_from_args(args), z, p, batch)
        (dy,) = torch.autograd.grad(y.sum(), p, create_graph=True)
        ((-dy * R.double()).sum() + (y.view(-1) * ge.double()).sum()).backward()
        return {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}, p.grad

    args = dict(W.TINY_ARGS, derivative=True)
    torch.manual_seed(17)
    model = create_model(dict(args)).to("cuda")
    model.parameter_gradients = True
    model.force_gradient_order = 0
    model.one_pass_training = one_pass
    z, pos, batch = _ragged([22, 35, 9], seed=1300)
    R = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    ge = torch.tensor([0.7, -1.1, 0.4])
    pc = pos.cuda().requires_grad_(True)
    import warnings

    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*energy term's part.*")  # the analytic pass builds the position gradient: nothing to announce
        y, F = model(z.cuda(), pc, batch.cuda())
        loss = (F * R.cuda()).sum() + (y.view(-1) * ge.cuda()).sum()
        loss.backward()
    ref, ref_pos = _oracle_force_matching_grads(model, args, z, pos, batch, R, ge)
    pos_err = (pc.grad.cpu().double() - ref_pos).abs().max().item() / ref_pos.abs().max().item()
    errs = {}
    for k, p in model.named_parameters():
        if k in ref and ref[k].abs().max() > 0:
            assert p.grad is not None, k
            errs[k] = (p.grad.cpu().double() - ref[k]).abs().max().item() / ref[k].abs().max().item()
    worst = max(errs, key=errs.get)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/force_gradient_analytic_{'one' if one_pass else 'two'}_pass.json", "w") as fh:
        json.dump({"worst": [worst, errs[worst]], "position_gradient": pos_err, "errors": errs}, fh, indent=1)
    assert errs[worst] < REL, (worst, errs[worst])
    assert pos_err < REL, pos_err


@pytest.mark.parametrize("fixture", ["et_tiny_ref.pt", "et_tiny_vc_ref.pt"])
def test_et_analytic_force_term_gradients_match_specification(hip_lib, golden_dir, fixture):
    """Equivariant Transformer: tmdnet_force_param_grads (csrc/tn_et_hvp.hip + the schedule in tn_et_api.hip) against
    oracle/et_second_order.py in fp64 (pinned to autograd-of-autograd): every parameter, and H v - both distance filters with the
    neighbour embedding, and the vector cutoff with the key filter only."""
    from oracle import et_second_order as E2
    from oracle import et_torch as ET
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    z, pos, batch = g["z"], g["pos"], g["batch"]
    n_mol = int(batch.max()) + 1
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), None, None, n_mol, v.cuda(), want_hv=True)
    torch.cuda.synchronize()
    hp = ET.hparams_from_args(g["args"])
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    ref = E2.force_term(sd64, hp, z, pos.double(), batch, v.double())
    by_name = {id(p): k for k, p in model.named_parameters()}
    errs = {}
    for p, gr in grads.items():
        key = by_name[id(p)]
        r = ref["grads"].get(key)
        if r is None or r.abs().max() == 0:
            assert gr.abs().max().item() < 1e-6, key
            continue
        errs[key] = (gr.cpu().double() - r.reshape(gr.shape)).abs().max().item() / r.abs().max().item()
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()
    # the engine's intermediates against the host run of the same bodies, in schedule order (et_hvp_debug_tensor)
    from tests import et_hvp_host_mirror as EM
    from torchmdnet_amd import _C

    mir = EM.force_term_mirror(g["state_dict"], hp, z, pos, batch, v)
    L_, st = _C.lib(), model._engine
    rows = []
    for name in mir["order"]:
        r_ = mir["bufs"][name]
        n = L_.tmdnet_hvp_debug_tensor(st.handle, None, name.encode(), None, 0)
        if name in ("gdel", "gdel_t") and n != r_.numel():
            continue
        if n != r_.numel():
            rows.append((name, f"size {n} != {r_.numel()}"))
            continue
        out = torch.empty(r_.numel(), dtype=torch.float32, device="cuda")
        assert L_.tmdnet_hvp_debug_tensor(st.handle, None, name.encode(), C.c_void_p(out.data_ptr()), out.numel()) == 0
        torch.cuda.synchronize()
        o, r = out.cpu(), r_.reshape(-1).clone()
        if name in mir["never"]:  # the self pair's row of every per-pair block is never written (nor read)
            nb, P1_ = mir["never"][name]
            keep = torch.ones(nb, P1_, r.numel() // (nb * P1_), dtype=torch.bool)
            keep[:, P1_ - 1] = False
            o, r = o[keep.reshape(-1)], r[keep.reshape(-1)]
        err = (o - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        rows.append((name, err if torch.isfinite(o).all() else float("inf")))
    first_bad = next(((n, e) for n, e in rows if not (isinstance(e, float) and e < 1e-3)), None)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_{fixture[:-3]}.json", "w") as fh:
        json.dump({"case": fixture, "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs,
                   "first_bad_buffer": first_bad, "buffers": rows}, fh, indent=1)
    assert first_bad is None, first_bad
    assert len(errs) >= 30
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert not bad, bad
    assert hv_err < REL, hv_err


@pytest.mark.parametrize("fixture", ["tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_tn2_analytic_force_term_gradients_match_specification(hip_lib, golden_dir, fixture):
    """TensorNet2 + ScalarPlusWeightedCoulomb: tmdnet_force_param_grads (csrc/tn_tn2_hvp.hip + the T2 statements of the schedule in
    tn_hvp_api.hip) against oracle/tn2_second_order.py in fp64 (pinned to autograd-of-autograd): every parameter - the charge heads
    and the three blocks of the first edge layer included - and H v with the Coulomb term's own pair geometry; total charges with the
    all-to-all sum, and the reaction-field branch in a periodic box.  The intermediates are walked against the host run of the same
    bodies (tests/tn2_hvp_host_mirror.py): per-edge tensors ("l*.pre1" ... "g1") assume the same CSR edge order on both sides - if
    every per-edge tensor differs and no per-atom one does, look there first."""
    from oracle import tn2_second_order as N2
    from oracle import tn2_torch as T2
    from tests import tn2_hvp_host_mirror as M2
    from torchmdnet_amd import _C
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    z, pos, batch = g["z"], g["pos"], g["batch"]
    q = g["q"] if g.get("q") is not None else None
    box = g["box"] if g.get("box") is not None else None
    n_mol = int(batch.max()) + 1
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), None if box is None else box.cuda(),
                                                     None if q is None else q.cuda(), n_mol, v.cuda(), want_hv=True)
    torch.cuda.synchronize()
    hp = T2.hparams_from_args(g["args"])
    mir = M2.force_term_mirror(g["state_dict"], hp, z, pos, batch, v, box=box, q=q)
    L, st = _C.lib(), model._engine
    rows = []
    for name in mir["order"]:
        ref = mir["bufs"][name]
        n = L.tmdnet_hvp_debug_tensor(st.handle, None, name.encode(), None, 0)
        if n != ref.numel():
            rows.append((name, f"size {n} != {ref.numel()}"))
            continue
        out = torch.empty(ref.numel(), dtype=torch.float32, device="cuda")
        assert L.tmdnet_hvp_debug_tensor(st.handle, None, name.encode(), C.c_void_p(out.data_ptr()), out.numel()) == 0
        torch.cuda.synchronize()
        o, r = out.cpu(), ref.reshape(-1)
        if name in ("gAp", "gAp_t", "gdel", "gdel_t") and r.numel() != o.numel():
            continue
        err = (o - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        rows.append((name, err if torch.isfinite(o).all() else float("inf")))
    first_bad = next(((n, e) for n, e in rows if not (isinstance(e, float) and e < 1e-3)), None)
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    ref = N2.force_term(sd64, hp, z, pos.double(), batch, v.double(), box=None if box is None else box.double(),
                        q=None if q is None else q.double())
    refg = N2.state_dict_grads(ref, sd64, hp)
    by_name = {id(p): k for k, p in model.named_parameters()}
    errs = {}
    for p, gr in grads.items():
        key = by_name[id(p)]
        r = refg.get(key)
        if r is None or r.abs().max() == 0:
            assert gr.abs().max().item() < 1e-6, key
            continue
        errs[key] = (gr.cpu().double() - r.reshape(gr.shape)).abs().max().item() / r.abs().max().item()
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_{fixture[:-3]}.json", "w") as fh:
        json.dump({"case": fixture, "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs,
                   "first_bad_buffer": first_bad, "buffers": rows}, fh, indent=1)
    assert first_bad is None, first_bad
    missing = {k for k, t in refg.items() if t.abs().max() > 0} - set(errs)
    assert not missing, missing
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert not bad, bad
    assert hv_err < REL, hv_err


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_engine_second_order_pass_equals_the_reference_double_backward(hip_lib, golden_dir, fixture):
    """The engine's analytic pass directly against what the UNMODIFIED reference's second autograd pass gave for the same weights,
    inputs and direction v (tests/golden/second_order_ref.pt, fp64, made by oracle/make_golden_second_order.py): H v and every
    parameter the reference has a gradient for, at the north-star bound."""
    from torchmdnet_amd.models.model import create_model

    ref = torch.load(os.path.join(golden_dir, "second_order_ref.pt"))[fixture]
    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    z, pos, batch = g["z"], g["pos"], g["batch"]
    q = g["q"] if g.get("q") is not None else None
    n_mol = int(batch.max()) + 1
    box = g["box"] if g.get("box") is not None else None
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), None if box is None else box.cuda(),
                                                     None if q is None else q.cuda(), n_mol, ref["v"].float().cuda(), want_hv=True)
    torch.cuda.synchronize()
    by_name = {id(p): k for k, p in model.named_parameters()}
    mine = {by_name[id(p)]: t.cpu().double() for p, t in grads.items()}
    errs = {}
    for k, r in ref["grads"].items():
        if r.abs().max() == 0:
            continue
        assert k in mine, k
        errs[k] = (mine[k].reshape(r.shape) - r).abs().max().item() / r.abs().max().item()
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_vs_reference_{fixture[:-3]}.json", "w") as fh:
        json.dump({"case": fixture, "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs}, fh, indent=1)
    assert len(errs) >= 30
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert not bad, bad
    assert hv_err < REL, hv_err


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_seeded_second_order_pass_matches_specification(hip_lib, golden_dir, fixture):
    """tmdnet_loss_param_grads: with ge = d loss / d E the pass returns the gradient of  S = v . d(sum E)/d pos - sum_m ge_m E_m  in
    every parameter and in the positions (one-pass training), against the specifications in fp64 (pinned to autograd of that scalar,
    tests/test_oracle.py)."""
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in g["state_dict"].items()}
    z, pos, batch = g["z"], g["pos"], g["batch"]
    q = g["q"] if g.get("q") is not None else None
    box = g["box"] if g.get("box") is not None else None
    n_mol = int(batch.max()) + 1
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    ge = torch.randn(n_mol, generator=torch.Generator().manual_seed(4))
    d = lambda t: None if t is None else t.double()
    c = lambda t: None if t is None else t.cuda()
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), c(box), c(q), n_mol, v.cuda(), want_hv=True, ge=ge.cuda())
    torch.cuda.synchronize()
    if fixture.startswith("et_"):
        from oracle import et_second_order as E2
        from oracle import et_torch as ET

        ref = E2.force_term(sd64, ET.hparams_from_args(g["args"]), z, pos.double(), batch, v.double(), ge=ge.double())
        refg = ref["grads"]
    elif fixture.startswith("tn2_"):
        from oracle import tn2_second_order as N2
        from oracle import tn2_torch as T2

        hp = T2.hparams_from_args(g["args"])
        ref = N2.force_term(sd64, hp, z, pos.double(), batch, v.double(), box=d(box), q=d(q), ge=ge.double())
        refg = N2.state_dict_grads(ref, sd64, hp)
    else:
        from oracle import tensornet_second_order as S2
        from oracle import tensornet_torch as T

        hp = T.hparams_from_args(g["args"])
        ref = S2.force_term(sd64, hp, z, pos.double(), batch, v.double(), q=d(q), ge=ge.double())
        refg = S2.state_dict_grads(ref["ent"], sd64, hp)
    by_name = {id(p): k for k, p in model.named_parameters()}
    mine = {by_name[id(p)]: t.cpu().double() for p, t in grads.items()}
    errs = {}
    for k, r in refg.items():
        if r.abs().max() == 0:
            continue
        assert k in mine, k
        errs[k] = (mine[k].reshape(r.shape) - r).abs().max().item() / r.abs().max().item()
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_seeded_{fixture[:-3]}.json", "w") as fh:
        json.dump({"case": fixture, "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs}, fh, indent=1)
    assert len(errs) >= 30
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert not bad, bad
    assert hv_err < REL, hv_err


@pytest.mark.parametrize("name,extra,sizes,charges", [
    ("so3-one-layer-no-total-charge", dict(equivariance_invariance_group="SO(3)", num_layers=1, q_weights=[1.0, 0.5]), [9, 14], False),
    ("three-layers-small-q", dict(num_layers=3, q_dim=4, q_weights=[1.0, 0.5, 2.0, 1.5]), [7, 12, 1, 5], True),
    ("reaction-field-no-box", dict(coulomb_cutoff=4.0), [11, 8], True),
])
def test_tn2_analytic_pass_on_other_configurations(hip_lib, name, extra, sizes, charges):
    """TensorNet2 branches the two fixtures do not reach (SO(3), one / three layers, no total charge, a single-atom molecule, the
    reaction-field term without a box), random-init, with the energy seed: engine vs specification, parameters and position gradient."""
    from oracle import tn2_second_order as N2
    from oracle import tn2_torch as T2
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=8, q_weights=[1.0, 0.5, 2.0], derivative=True)
    args.update(extra)
    torch.manual_seed(29)
    model = create_model(dict(args))
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "charge_predict" in k and p.dim() > 1:
                p.mul_(3.0)
    model = model.to("cuda")
    z, pos, batch = _ragged(sizes, seed=1700)
    B = len(sizes)
    q = torch.tensor([float(m % 3 - 1) for m in range(B)]) if charges else None
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    ge = torch.randn(B, generator=torch.Generator().manual_seed(6))
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), None, None if q is None else q.cuda(), B, v.cuda(),
                                                     want_hv=True, ge=ge.cuda())
    torch.cuda.synchronize()
    sd64 = {k: (t.detach().cpu().double() if t.is_floating_point() else t.detach().cpu()) for k, t in model.state_dict().items()}
    hp = T2.hparams_from_args(args)
    ref = N2.force_term(sd64, hp, z, pos.double(), batch, v.double(), q=None if q is None else q.double(), ge=ge.double())
    refg = N2.state_dict_grads(ref, sd64, hp)
    by_name = {id(p): k for k, p in model.named_parameters()}
    mine = {by_name[id(p)]: t.cpu().double() for p, t in grads.items()}
    errs = {k: (mine[k].reshape(r.shape) - r).abs().max().item() / r.abs().max().item() for k, r in refg.items() if r.abs().max() > 0}
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_tn2_{name}.json", "w") as fh:
        json.dump({"case": name, "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs}, fh, indent=1)
    bad = {k: e for k, e in errs.items() if not e < 5 * REL}  # random-init heads: fp32 conditioning (host run of the bodies: 2e-4 bound)
    assert not bad, bad
    assert hv_err < 5 * REL, hv_err
def test_analytic_force_term_gradients_in_a_periodic_box(hip_lib, golden_dir):
    """The second-order pass on a periodic system (triclinic box, minimum-image pairs from the brute-force graph): parameters and
    H v against the specification in fp64 (pinned to autograd-of-autograd with the same box, tests/test_oracle.py)."""
    from oracle import tensornet_second_order as S2
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    tiny = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    f = torch.load(os.path.join(golden_dir, "tiny_pbc_ref.pt"))
    model = create_model(dict(f["args"]))
    model.load_state_dict(tiny["state_dict"])
    model = model.to("cuda")
    z, pos, batch, box = f["z"], f["pos"], f["batch"], f["box"]
    n_mol = int(batch.max()) + 1
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(7))
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), box.cuda(), None, n_mol, v.cuda(), want_hv=True)
    torch.cuda.synchronize()
    sd64 = T.cast_state_dict(tiny["state_dict"], torch.float64)
    hp = T.hparams_from_args(f["args"])
    ref = S2.force_term(sd64, hp, z, pos.double(), batch, v.double(), box=box.double())
    refg = S2.state_dict_grads(ref["ent"], sd64, hp)
    by_name = {id(p): k for k, p in model.named_parameters()}
    errs = {}
    for p, g in grads.items():
        r = refg[by_name[id(p)]].reshape(g.shape)
        if r.abs().max() > 0:
            errs[by_name[id(p)]] = (g.cpu().double() - r).abs().max().item() / r.abs().max().item()
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / ref["Hv"].abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/hvp_periodic.json", "w") as fh:
        json.dump({"hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs}, fh, indent=1)
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert len(errs) >= 30 and not bad, bad
    assert hv_err < REL, hv_err


@pytest.mark.parametrize("name,extra,sizes", [
    ("values-only-no-neighbour-embedding", dict(distance_influence="values", neighbor_embedding=False), [9, 14, 1]),
    ("no-filters", dict(distance_influence="none"), [12, 5]),
    ("keys-vector-cutoff-three-layers", dict(distance_influence="keys", vector_cutoff=True, num_layers=3), [7, 11, 2]),
])
def test_et_analytic_pass_on_other_configurations(hip_lib, name, extra, sizes):
    """Equivariant Transformer branches the two fixtures do not reach (no neighbour embedding, value filter only, no distance filter at
    all, a single-atom molecule, three layers), random-init, with the energy seed: engine vs specification in fp64."""
    from oracle import et_second_order as E2
    from oracle import et_torch as ET
    from torchmdnet_amd.models.model import create_model

    args = dict(W.ET_TINY_ARGS, derivative=True, **extra)
    torch.manual_seed(23)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = _ragged(sizes, seed=900)
    B = len(sizes)
    v = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    ge = torch.randn(B, generator=torch.Generator().manual_seed(6))
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), None, None, B, v.cuda(), want_hv=True, ge=ge.cuda())
    torch.cuda.synchronize()
    sd64 = {k: (t.detach().cpu().double() if t.is_floating_point() else t.detach().cpu()) for k, t in model.state_dict().items()}
    ref = E2.force_term(sd64, ET.hparams_from_args(args), z, pos.double(), batch, v.double(), ge=ge.double())
    by_name = {id(p): k for k, p in model.named_parameters()}
    mine = {by_name[id(p)]: t.cpu().double() for p, t in grads.items()}
    errs = {k: (mine[k].reshape(r.shape) - r).abs().max().item() / r.abs().max().item() for k, r in ref["grads"].items() if r.abs().max() > 0}
    hv_err = (hv.cpu().double() - ref["Hv"]).abs().max().item() / max(ref["Hv"].abs().max().item(), 1e-30)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/hvp_et_{name}.json", "w") as fh:
        json.dump({"case": name, "hv_error": hv_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs}, fh, indent=1)
    bad = {k: e for k, e in errs.items() if not e < 1e-3}  # random-init heads reach 1e3: fp32 conditioning (host run of the bodies: 2e-4 there)
    assert len(errs) >= 20 and not bad, bad
    assert hv_err < 1e-3, hv_err


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_one_pass_gradient_equals_the_reference_backward_of_an_energy_and_force_loss(hip_lib, golden_dir, fixture):
    """tmdnet_loss_param_grads directly against ONE backward of the unmodified reference for loss = sum_m ge_m E_m + sum_i v_i . F_i
    (tests/golden/second_order_ref.pt): d loss / d theta = - grads, d loss / d pos = - hv."""
    from torchmdnet_amd.models.model import create_model

    ref = torch.load(os.path.join(golden_dir, "second_order_ref.pt"))[fixture]
    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    z, pos, batch = g["z"], g["pos"], g["batch"]
    q = g["q"] if g.get("q") is not None else None
    box = g["box"] if g.get("box") is not None else None
    c = lambda t: None if t is None else t.cuda()
    n_mol = int(batch.max()) + 1
    grads, hv = model.force_term_parameter_gradients(z.cuda(), pos.cuda(), batch.cuda(), c(box), c(q), n_mol, ref["v"].float().cuda(), want_hv=True,
                                                     ge=ref["ge"].float().cuda())
    torch.cuda.synchronize()
    by_name = {id(p): k for k, p in model.named_parameters()}
    mine = {by_name[id(p)]: -t.cpu().double() for p, t in grads.items()}
    errs = {k: (mine[k].reshape(r.shape) - r).abs().max().item() / r.abs().max().item() for k, r in ref["loss_grads"].items()
            if r.abs().max() > 0}
    pos_err = (-hv.cpu().double() - ref["loss_pos_grad"]).abs().max().item() / ref["loss_pos_grad"].abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/one_pass_vs_reference_{fixture[:-3]}.json", "w") as fh:
        json.dump({"case": fixture, "position_gradient": pos_err, "worst_param": max(errs.items(), key=lambda kv: kv[1]), "param_errors": errs}, fh,
                  indent=1)
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert len(errs) >= 30 and not bad, bad
    assert pos_err < REL, pos_err


@pytest.mark.parametrize("fixture", ["tiny_ref.pt", "et_tiny_ref.pt", "et_tiny_vc_ref.pt", "tn2_tiny_ref.pt", "tn2_tiny_rf_ref.pt"])
def test_default_training_backward_equals_the_reference_backward(hip_lib, golden_dir, fixture):
    """The path a training script takes with NO switch touched except `parameter_gradients` (reference: model.py:618-628 with
    create_graph=self.training, module.py:200-290): `loss(E, F).backward()` through autograd.  force_gradient_order = None must
    mean the analytic pass for TensorNet, the Equivariant Transformer AND TensorNet2, in one seeded pass: .grad of every parameter
    and pos.grad against ONE backward of the unmodified reference (tests/golden/second_order_ref.pt) at 1e-4, and no
    truncated-position-gradient warning."""
    import warnings
    from torchmdnet_amd.models.model import create_model

    ref = torch.load(os.path.join(golden_dir, "second_order_ref.pt"))[fixture]
    g = torch.load(os.path.join(golden_dir, fixture))
    model = create_model(dict(g["args"], derivative=True))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    model.parameter_gradients = True
    assert model.force_gradient_order is None and model.one_pass_training
    c = lambda t: None if t is None else t.cuda()
    pos = g["pos"].cuda().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        E, F = model(g["z"].cuda(), pos, g["batch"].cuda(), box=c(g.get("box")), q=c(g.get("q")))
        loss = (E.view(-1) * ref["ge"].float().cuda()).sum() + (F * ref["v"].float().cuda()).sum()
        loss.backward()
    errs = {}
    for k, p in model.named_parameters():
        r = ref["loss_grads"].get(k)
        if r is not None and r.abs().max() > 0:
            assert p.grad is not None, k
            errs[k] = (p.grad.cpu().double().reshape(r.shape) - r).abs().max().item() / r.abs().max().item()
    pos_err = (pos.grad.cpu().double() - ref["loss_pos_grad"]).abs().max().item() / ref["loss_pos_grad"].abs().max().item()
    bad = {k: e for k, e in errs.items() if not e < REL}
    assert len(errs) >= 30 and not bad, bad
    assert pos_err < REL, pos_err
