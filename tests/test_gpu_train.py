"""-m gpu: first-order parameter gradients of TensorNet + Scalar (tmdnet_energy_param_grads, csrc/tn_train.hip + the hooks in the
reverse pass of csrc/tn_api.hip) against autograd over the oracle (oracle/tensornet_torch.py restates reference tensornet.py:
543-619, 729-814, 384-398 and output_modules.py:108-117 in plain torch, so `energy(sd, ...)` is differentiable in every weight).
The reference gets these gradients from autograd; the bound is the north-star 1e-4 relative to each tensor's largest entry."""
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


def _oracle_grads(model, args, z, pos, batch, q, ge, dtype=torch.float64):
    from oracle import tensornet_torch as T

    sd = {k: v.detach().cpu().to(dtype).requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    y = T.energy(sd, T.hparams_from_args(args), z, pos.to(dtype), batch, q=None if q is None else q.to(dtype))
    (y.view(-1) * ge.to(dtype)).sum().backward()
    return y.detach().view(-1), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}


@pytest.mark.parametrize("name,extra,sizes,charges", [
    ("tiny-o3-charges", dict(), [18, 30, 4, 11, 1], True),
    ("so3", dict(equivariance_invariance_group="SO(3)"), [9, 21, 14], False),
    ("wide", dict(embedding_dimension=128, num_rbf=32, num_layers=2), [40, 33, 64], True),
    ("one-layer-lower-cutoff", dict(num_layers=1, cutoff_lower=0.8, cutoff_upper=4.5), [25, 12], False),
    ("odd-widths", dict(embedding_dimension=64, num_rbf=50, max_z=100, num_layers=3), [64, 64, 37, 64, 2], True),
])
def test_parameter_gradients_match_oracle_autograd(hip_lib, name, extra, sizes, charges):
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, **extra)
    torch.manual_seed(11)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = _ragged(sizes, seed=500)
    B = len(sizes)
    q = torch.tensor([float(m % 3 - 1) for m in range(B)]) if charges else None
    ge = torch.linspace(-1.0, 1.5, B)  # a different seed per molecule: d loss / d E_m
    E, grads = model.parameter_gradients_of(z.cuda(), pos.cuda(), batch.cuda(), None, None if q is None else q.cuda(), B, ge.cuda())
    Er, ref = _oracle_grads(model, args, z, pos, batch, q, ge)
    assert (E.cpu().double() - Er).abs().max() / Er.abs().max() < REL
    by_name = {id(p): k for k, p in model.named_parameters()}
    seen, bad = set(), {}
    for p, g in grads.items():
        key = by_name[id(p)]
        seen.add(key)
        r = ref[key].reshape(g.shape)
        err = (g.cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        if not err < REL:
            bad[key] = err
    assert not bad, (name, bad)
    names = {k for k, _ in model.named_parameters()}  # (mean / std / the radial basis are buffers, as in the reference)
    missing = [k for k in ref if k in names and k not in seen and ref[k].abs().max() > 0]
    assert not missing, missing  # every weight the energy depends on has a gradient
    # deterministic: the same call gives the same bits
    E2, grads2 = model.parameter_gradients_of(z.cuda(), pos.cuda(), batch.cuda(), None, None if q is None else q.cuda(), B, ge.cuda())
    assert torch.equal(E, E2) and all(torch.equal(grads[p], grads2[p]) for p in grads)


@pytest.mark.parametrize("name,extra", [
    ("both-filters-neighbour-embedding", dict()),
    ("vector-cutoff-keys-only", dict(vector_cutoff=True, distance_influence="keys")),
    ("values-only-no-neighbour-embedding", dict(distance_influence="values", neighbor_embedding=False, num_layers=3)),
    ("no-filters", dict(distance_influence="none")),
])
def test_et_parameter_gradients_match_oracle_autograd(hip_lib, name, extra):
    """Equivariant Transformer + EquivariantScalar (reference torchmd_et.py:188-426, models/utils.py:83-117, 583-655): the same
    pass for its weights - attention projections, distance filters (per-pair adjoint rows from the attention sweep), neighbour
    embedding, the two gated blocks of the head - against autograd over oracle/et_torch.py in fp64"""
    from oracle import et_torch as T
    from torchmdnet_amd.models.model import create_model

    args = dict(W.ET_TINY_ARGS, **extra)
    torch.manual_seed(13)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = _ragged([17, 29, 1, 8], seed=700)
    B = 4
    ge = torch.tensor([0.9, -0.4, 1.3, 0.2])
    E, grads = model.parameter_gradients_of(z.cuda(), pos.cuda(), batch.cuda(), None, None, B, ge.cuda())
    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    y = T.energy(sd, T.hparams_from_args(args), z, pos.double(), batch)
    (y.view(-1) * ge.double()).sum().backward()
    assert (E.cpu().double() - y.detach().view(-1)).abs().max() / y.detach().abs().max() < REL
    by_name = {id(p): k for k, p in model.named_parameters()}
    seen, bad = set(), {}
    for p, g in grads.items():
        key = by_name[id(p)]
        seen.add(key)
        r = sd[key].grad
        r = torch.zeros_like(sd[key]) if r is None else r
        err = (g.cpu().double() - r.reshape(g.shape)).abs().max().item() / max(r.abs().max().item(), 1e-30)
        if not err < REL and r.abs().max() > 0:
            bad[key] = err
        if r.abs().max() == 0:
            assert g.abs().max() == 0, key
    assert not bad, (name, bad)
    names = {k for k, _ in model.named_parameters()}
    missing = [k for k in names if k not in seen and sd[k].grad is not None and sd[k].grad.abs().max() > 0]
    assert not missing, missing


@pytest.mark.parametrize("name,extra,charges", [
    ("o3-charges", dict(), True),
    ("so3-cutoff-coulomb", dict(equivariance_invariance_group="SO(3)", coulomb_cutoff=6.0), False),
    ("three-layers", dict(num_layers=3, embedding_dimension=64), True),
])
def test_tn2_parameter_gradients_match_oracle_autograd(hip_lib, name, extra, charges):
    """TensorNet2 + ScalarPlusWeightedCoulomb (reference tensornet2.py:49-157, 372-626, output_modules.py:440-606): TensorNet's
    products plus the per-edge MLP with the charge channels of both endpoints (its reverse pass already goes through the MLP per
    edge), the ChargePredict heads with their per-molecule equilibration, and the Coulomb term's path to the charges"""
    from oracle import tn2_torch as T2
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=4, q_weights=[1.0, 0.5, 2.0], **extra)
    if "num_layers" in extra:
        args["q_weights"] = [1.0, 0.5, 2.0, 0.7]
    torch.manual_seed(29)
    model = create_model(dict(args), mean=torch.tensor(0.5), std=torch.tensor(2.0)).to("cuda")
    z, pos, batch = _ragged([19, 27, 2, 11], seed=2100)
    B = 4
    q = torch.tensor([1.0, -1.0, 0.0, 2.0]) if charges else torch.zeros(B)
    ge = torch.tensor([0.7, -0.5, 1.2, 0.3])
    E, grads = model.parameter_gradients_of(z.cuda(), pos.cuda(), batch.cuda(), None, q.cuda(), B, ge.cuda())
    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    y = T2.energy(sd, T2.hparams_from_args(args), z, pos.double(), batch, q=q.double())
    (y.view(-1) * ge.double()).sum().backward()
    assert (E.cpu().double() - y.detach().view(-1)).abs().max() / y.detach().abs().max() < REL
    by_name = {id(p): k for k, p in model.named_parameters()}
    seen, bad = set(), {}
    for p, g in grads.items():
        key = by_name[id(p)]
        seen.add(key)
        r = sd[key].grad
        if r is None or r.abs().max() == 0:
            continue
        err = (g.cpu().double() - r.reshape(g.shape)).abs().max().item() / r.abs().max().item()
        if not err < REL:
            bad[key] = err
    assert not bad, (name, bad)
    names = {k for k, _ in model.named_parameters()}
    missing = [k for k in names if k not in seen and sd[k].grad is not None and sd[k].grad.abs().max() > 0]
    assert not missing, missing


def test_parameter_gradients_periodic_box_standardisation_atomref(hip_lib):
    """one periodic (triclinic) box with minimum-image pairs, mean / std standardisation and an Atomref prior: std scales every
    gradient, the prior's table gets d loss / d atomref[t] = sum of the seeds of the atoms of species t"""
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, prior_model="Atomref", prior_args={"max_z": 20}, cutoff_upper=4.0, max_num_neighbors=96)
    torch.manual_seed(23)
    model = create_model(dict(args), mean=torch.tensor(1.75), std=torch.tensor(0.375))
    table = torch.randn(20, 1) * 3.0
    with torch.no_grad():
        model.prior_model[0].atomref.weight.copy_(table)
    model = model.to("cuda")
    z, pos, box = W.water_box(n_side=3, spacing=3.1)  # 81 atoms, L = 9.3 A > 2 rc
    z = z % 19 + 1
    box = box.clone()
    box[1, 0], box[2, 0], box[2, 1] = 0.9, -0.6, 0.4  # lower-triangular triclinic cell
    batch = torch.zeros_like(z)
    ge = torch.tensor([0.8])
    E, grads = model.parameter_gradients_of(z.cuda(), pos.cuda(), batch.cuda(), box.cuda(), None, 1, ge.cuda())
    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    at = table.double().requires_grad_(True)
    y = T.energy(sd, T.hparams_from_args(args), z, pos.double(), batch, box=box.double(), atomref=at)
    (y.view(-1) * ge.double()).sum().backward()
    assert (E.cpu().double() - y.detach().view(-1)).abs().max() / y.detach().abs().max() < REL
    by_name = {id(p): k for k, p in model.named_parameters()}
    bad = {}
    for p, g in grads.items():
        key = by_name[id(p)]
        r = at.grad if key.endswith("atomref.weight") else sd[key].grad
        err = (g.cpu().double() - r.reshape(g.shape)).abs().max().item() / max(r.abs().max().item(), 1e-30)
        if not err < REL:
            bad[key] = err
    assert not bad, bad
    assert any(by_name[id(p)].endswith("atomref.weight") for p in grads)


def test_energy_only_training_through_autograd(hip_lib):
    """parameter_gradients=True: `loss(y).backward()` fills .grad like the reference's autograd does; a few optimizer steps on
    a toy regression lower the loss, and the inference schedule afterwards evaluates the trained weights"""
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, derivative=False)
    torch.manual_seed(5)
    model = create_model(dict(args)).to("cuda")
    model.parameter_gradients = True
    z, pos, batch = _ragged([20, 31, 7, 16], seed=900)
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    target = torch.tensor([[0.3], [-0.2], [0.05], [0.4]]).cuda()
    y, _ = model(zc, pc, bc)
    loss = ((y - target) ** 2).sum()
    loss.backward()
    # .grad against autograd over the oracle with the same seeds 2 (y - target)
    ge = (2 * (y.detach() - target)).view(-1).cpu()
    _, ref = _oracle_grads(model, args, z, pos, batch, None, ge)
    for k, p in model.named_parameters():
        if k in ref and ref[k].abs().max() > 0:
            assert p.grad is not None, k
            err = (p.grad.cpu().double() - ref[k]).abs().max().item() / ref[k].abs().max().item()
            assert err < REL, (k, err)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        y, _ = model(zc, pc, bc)
        loss = ((y - target) ** 2).sum()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], losses
    # back on the inference schedule with the trained weights: same energies as the oracle
    model.parameter_gradients = False
    with torch.no_grad():
        y2, _ = model(zc, pc, bc)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    Er = T.energy(sd, T.hparams_from_args(args), z, pos, batch).detach()
    assert (y2.cpu() - Er).abs().max() / Er.abs().max() < REL


def _oracle_force_matching_grads(model, args, z, pos, batch, R, ge):
    """d/d theta of  sum_i R_i . F_i + sum_m ge_m E_m  by double backward over the oracle in fp64"""
    from oracle import tensornet_torch as T

    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    p = pos.double().clone().requires_grad_(True)
    y = T.energy(sd, T.hparams_from_args(args), z, p, batch)
    (dy,) = torch.autograd.grad(y.sum(), p, create_graph=True)
    loss = (-dy * R.double()).sum() + (y.view(-1) * ge.double()).sum()
    loss.backward()
    return {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}


@pytest.mark.parametrize("order,bound", [(2, 2e-3), (4, 1e-3)])
def test_force_matching_gradients_by_central_difference(hip_lib, order, bound):
    """derivative=True + parameter_gradients=True: the force output carries a graph to the weights whose backward is a central
    difference of the exact parameter gradient along d loss / d F.  Against the analytic double backward of the oracle (fp64)
    the error is dominated by fp32 rounding / step size: bounds stated here, measured values written to gpurun_out/."""
    import json, os
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, derivative=True)
    torch.manual_seed(17)
    model = create_model(dict(args)).to("cuda")
    model.parameter_gradients = True
    model.force_gradient_order = order
    z, pos, batch = _ragged([22, 35, 9], seed=1300)
    R = torch.randn(pos.shape, generator=torch.Generator().manual_seed(3))
    ge = torch.tensor([0.7, -1.1, 0.4])
    y, F = model(z.cuda(), pos.cuda(), batch.cuda())
    # the training step's inference-schedule call ran without the radial tables, and the switches are back afterwards
    assert model.engine_info("edge_table_min_pairs") < 1e9 and model.engine_info("embed_rb_min_atoms") < 1e9
    loss = (F * R.cuda()).sum() + (y.view(-1) * ge.cuda()).sum()
    loss.backward()
    assert model.engine_info("edge_table_min_pairs") < 1e9
    ref = _oracle_force_matching_grads(model, args, z, pos, batch, R, ge)
    errs = {}
    for k, p in model.named_parameters():
        if k in ref and ref[k].abs().max() > 0:
            assert p.grad is not None, k
            errs[k] = (p.grad.cpu().double() - ref[k]).abs().max().item() / ref[k].abs().max().item()
    worst = max(errs, key=errs.get)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/force_gradient_fd_order{order}.json", "w") as fh:
        json.dump({"order": order, "step": model.force_gradient_step or (0.02 if order == 4 else 0.005), "worst": [worst, errs[worst]], "errors": errs}, fh, indent=1)
    assert errs[worst] < bound, (worst, errs[worst])


def test_et_training_through_autograd(hip_lib):
    """Equivariant Transformer, derivative=True: energies and forces carry graphs to the weights (the analytic one-pass default);
    .grad against the oracle's fp64 double backward at 1e-4, and a few Adam steps lower the loss"""
    from oracle import et_torch as T
    from torchmdnet_amd.models.model import create_model

    args = dict(W.ET_TINY_ARGS, derivative=True)
    torch.manual_seed(19)
    model = create_model(dict(args)).to("cuda")
    model.parameter_gradients = True
    z, pos, batch = _ragged([15, 24, 6], seed=1500)
    R = torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    ge = torch.tensor([0.6, -0.9, 0.3])
    y, F = model(z.cuda(), pos.cuda(), batch.cuda())
    ((F * R.cuda()).sum() + (y.view(-1) * ge.cuda()).sum()).backward()
    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    p = pos.double().clone().requires_grad_(True)
    yr = T.energy(sd, T.hparams_from_args(args), z, p, batch)
    (dy,) = torch.autograd.grad(yr.sum(), p, create_graph=True)
    ((-dy * R.double()).sum() + (yr.view(-1) * ge.double()).sum()).backward()
    worst = 0.0
    for k, prm in model.named_parameters():
        r = sd[k].grad
        if r is not None and r.abs().max() > 0:
            assert prm.grad is not None, k
            worst = max(worst, (prm.grad.cpu().double() - r).abs().max().item() / r.abs().max().item())
    assert worst < 1e-4, worst
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    tgt_e, tgt_f = torch.tensor([[0.2], [-0.1], [0.3]]).cuda(), torch.zeros_like(pos).cuda()
    losses = []
    for _ in range(10):
        opt.zero_grad()
        y, F = model(z.cuda(), pos.cuda(), batch.cuda())
        loss = ((y - tgt_e) ** 2).sum() + 0.1 * ((F - tgt_f) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0], losses


def test_two_forwards_before_one_backward(hip_lib):
    """the forward half keeps its activations in the model's workspaces; a second forward (or any other engine call) before the
    backward takes them away, and the backward of the first then repeats the pass from its inputs: same gradients either way"""
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, derivative=False)
    torch.manual_seed(31)
    model = create_model(dict(args)).to("cuda")
    model.parameter_gradients = True
    za, pa, ba = _ragged([14, 9], seed=40)
    zb, pb, bb = _ragged([21, 5, 12], seed=60)
    ya, _ = model(za.cuda(), pa.cuda(), ba.cuda())
    yb, _ = model(zb.cuda(), pb.cuda(), bb.cuda())
    with torch.no_grad():
        model.parameter_gradients = False
        model(za.cuda(), pa.cuda(), ba.cuda())  # an inference call in between as well
        model.parameter_gradients = True
    (ya.sum() * 0.5 - yb.sum()).backward()
    got = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    _, ra = _oracle_grads(model, args, za, pa, ba, None, torch.full((2,), 0.5))
    _, rb = _oracle_grads(model, args, zb, pb, bb, None, torch.full((3,), -1.0))
    for k, g in got.items():
        r = ra[k] + rb[k]
        if r.abs().max() > 0:
            assert (g.cpu().double() - r).abs().max().item() / r.abs().max().item() < REL, k


@pytest.mark.parametrize("arch", ["tensornet", "tensornet2"])
def test_parameter_update_stays_on_the_device(hip_lib, arch):
    """An in-place optimizer step on CUDA parameters reaches the engine through tmdnet_update_params_device (a gather kernel
    from the caller's tensors + the image kernels; nothing but a pointer table crosses PCIe): the result must be BIT-IDENTICAL to
    a fresh handle that uploaded the same weights through the host."""
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, embedding_dimension=64, num_rbf=16, num_layers=2)
    if arch == "tensornet2":
        args.update(model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=4, q_weights=[1.0, 0.5, 2.0])
    torch.manual_seed(3)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = (t.cuda() for t in _ragged([17, 30, 9], seed=77))
    q = torch.tensor([0.0, 1.0, -1.0]).cuda() if arch == "tensornet2" else None
    kw = dict(q=q) if q is not None else {}
    model(z, pos, batch, **kw)  # full upload
    assert model._engine.device_updates == 0
    g = torch.Generator(device="cuda").manual_seed(5)
    for step in range(3):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01 * torch.randn(p.shape, device="cuda", generator=g) * p.abs().mean())
        E, F = model(z, pos, batch, **kw)
        assert model._engine.device_updates == step + 1
    fresh = create_model(dict(args)).to("cuda")
    fresh.load_state_dict(model.state_dict())
    Er, Fr = fresh(z, pos, batch, **kw)
    assert fresh._engine.device_updates == 0
    assert torch.equal(E, Er) and torch.equal(F, Fr)
    # a value the host reads (std) falls back to the full upload, and so does a replaced tensor object
    with torch.no_grad():
        model.std.mul_(2.0)
    n = model._engine.device_updates
    E2, _ = model(z, pos, batch, **kw)
    assert model._engine.device_updates == n and not torch.equal(E2, E)


def test_trained_model_evaluates_like_a_fresh_one_at_batch_scale_and_on_a_side_stream(hip_lib):
    """(1) the device-side update was queued on the caller's (non-blocking) stream while the radial tables were
    rebuilt with NULL-stream kernels - nothing ordered the rebuild after the gather; (2) the radial-basis embedding's images stayed
    off after the first device-side update, so a trained model took another schedule than a freshly loaded one.  Now: an update on a
    SIDE stream followed by a batch-scale evaluation (radial tables + radial-basis embedding: >= 1024 atoms, F = 128, K = 32) on
    that stream is bit-identical to a fresh handle with the same weights, and the embedding is on again."""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(4)
    model = create_model(dict(W.C2_ARGS)).to("cuda")
    z, pos, batch = (t.cuda() for t in W.synthetic_batch(n_mol=24, n_atoms=64))
    model(z, pos, batch)  # full upload; tables + images built
    assert model.engine_info("embed_rb") == 1.0
    side = torch.cuda.Stream()
    g = torch.Generator(device="cuda").manual_seed(6)
    for _ in range(2):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.02 * torch.randn(p.shape, device="cuda", generator=g) * p.abs().mean())
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            E, F = model(z, pos, batch)
        torch.cuda.current_stream().wait_stream(side)
    assert model._engine.device_updates == 2
    assert model.engine_info("embed_rb") == 1.0
    fresh = create_model(dict(W.C2_ARGS)).to("cuda")
    fresh.load_state_dict(model.state_dict())
    Er, Fr = fresh(z, pos, batch)
    torch.cuda.synchronize()
    assert torch.equal(E, Er) and torch.equal(F, Fr)


def test_force_loss_position_gradient_is_announced_as_truncated(hip_lib):
    """d loss / d pos THROUGH the forces is a second derivative in the positions.  The analytic pass (the default for every
    architecture) builds it (tests/test_gpu_hvp.py); the difference-quotient cross-check (order 2 / 4 on request) does not: pos always requires grad in derivative mode (the reference's side effect), so the backward cannot refuse - it
    warns once that pos.grad holds the energy term's part only, and the parameter gradients are unaffected."""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(2)
    model = create_model(dict(W.TINY_ARGS, derivative=True)).to("cuda")
    model.parameter_gradients = True
    model.force_gradient_order = 2
    z, pos, batch = (t.cuda() for t in _ragged([12, 20], seed=4))
    y, F = model(z, pos, batch)
    with pytest.warns(UserWarning, match="second derivative in the positions"):
        (F ** 2).sum().backward()
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)
