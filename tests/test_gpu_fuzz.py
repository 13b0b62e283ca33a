"""-m gpu: hyper-parameter sweep of the HIP path against the torch oracles (oracle/tensornet_torch.py, oracle/et_torch.py,
oracle/tn2_torch.py - each pinned to the unmodified reference in tests/test_oracle.py).  The fixed parity tests sit on the
shipped configurations (F = 32 / 128); this one walks the kernel variants behind other shapes: channel counts that are not a
multiple of 64 or of 32 (scalar-lane kernels, fp32-MFMA GEMMs instead of the split-bf16 ones), odd RBF counts, zero to three
layers, a lower cutoff, both symmetry groups, every ET option combination, molecules from 1 to 90 atoms in one batch, a
periodic box.  Random-init weights (seeded); tolerance 1e-4 relative in fp32 (the north-star bound)."""
import itertools

import pytest
import torch

from torchmdnet_amd import workloads as W

pytestmark = pytest.mark.gpu
REL = 1e-4


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _batch(sizes, seed, z_max):
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(seed + m, n_atoms=n)
        zs.append(torch.from_numpy(zz) % (z_max - 1) + 1)
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    return torch.cat(zs), torch.cat(ps), torch.cat(bs)


def _check(args, T, sizes, seed, box=None, q=None, hp_extra=None, options=None):
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(seed)
    model = create_model(dict(args)).to("cuda")
    for name, value in (options or {}).items():
        model.set_engine_option(name, value)
    z, pos, batch = _batch(sizes, 1000 + seed, int(args["max_z"]))
    kw = {}
    if box is not None:
        kw["box"] = box.cuda()
    if q is not None:
        kw["q"] = q.cuda()
    E, F = model(z.cuda(), pos.cuda(), batch.cuda(), **kw)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    okw = dict(box=box)
    if q is not None:
        okw["q"] = q
    Er, Fr = T.energy_and_forces(sd, T.hparams_from_args(args), z, pos, batch, **okw)
    assert torch.isfinite(E).all() and torch.isfinite(F).all()
    assert rel_err(E.cpu(), Er) < REL, (args, sizes)
    assert rel_err(F.cpu(), Fr) < REL, (args, sizes)
    # no net force on an isolated molecule / a periodic cell
    assert F.sum(0).abs().max().item() < 1e-3 * max(F.abs().max().item(), 1e-6)


TN_CASES = [
    # F,   K,  L, r_lo, r_hi, group,   sizes
    (16, 8, 1, 0.0, 4.0, "O(3)", [5, 9]),
    (24, 7, 2, 0.0, 5.0, "SO(3)", [1, 30, 2]),
    (48, 50, 0, 0.0, 5.0, "O(3)", [12, 33]),
    (64, 32, 3, 0.5, 4.5, "O(3)", [20, 64]),
    (96, 20, 2, 0.0, 6.0, "SO(3)", [90]),
    (160, 16, 1, 1.0, 5.0, "O(3)", [8, 40, 17]),
    (192, 33, 2, 0.0, 3.5, "O(3)", [64, 3]),
    (256, 32, 1, 0.0, 5.0, "O(3)", [25, 25]),
]


@pytest.mark.parametrize("case", TN_CASES, ids=lambda c: f"F{c[0]}-K{c[1]}-L{c[2]}-lo{c[3]}-{c[5]}")
@pytest.mark.parametrize("charged", [False, True])
def test_tensornet_shapes(hip_lib, case, charged):
    from oracle import tensornet_torch as T

    F, K, L, lo, hi, group, sizes = case
    args = dict(W.TINY_ARGS, embedding_dimension=F, num_rbf=K, num_layers=L, cutoff_lower=lo, cutoff_upper=hi,
                equivariance_invariance_group=group, max_z=12)
    q = torch.linspace(-1.0, 2.0, len(sizes)) if charged else None
    _check(args, T, sizes, seed=F + K + L, q=q)


def test_tensornet_periodic_shapes(hip_lib):
    from oracle import tensornet_torch as T

    box = torch.tensor([[11.0, 0.0, 0.0], [0.8, 10.5, 0.0], [-1.2, 1.1, 12.0]])
    for F, K, L in ((24, 9, 1), (96, 32, 2)):
        args = dict(W.TINY_ARGS, embedding_dimension=F, num_rbf=K, num_layers=L, cutoff_upper=4.5, max_z=12)
        _check(args, T, [70], seed=3 * F, box=box)


ET_OPTS = list(itertools.product(["both", "keys", "values", "none"], [False, True], [False, True]))


@pytest.mark.parametrize("influence,vector_cutoff,nbr_embedding", ET_OPTS,
                         ids=lambda v: str(v))
def test_et_options_and_shapes(hip_lib, influence, vector_cutoff, nbr_embedding):
    from oracle import et_torch as T

    k = ET_OPTS.index((influence, vector_cutoff, nbr_embedding))
    F, H, K, L = [(32, 4, 16, 2), (48, 3, 9, 1), (64, 8, 20, 3), (128, 8, 32, 1)][k % 4]
    args = dict(W.ET_TINY_ARGS, embedding_dimension=F, num_heads=H, num_rbf=K, num_layers=L, distance_influence=influence,
                vector_cutoff=vector_cutoff, neighbor_embedding=nbr_embedding, cutoff_lower=0.4 if k % 3 == 0 else 0.0,
                cutoff_upper=4.5 + 0.5 * (k % 3), max_z=12)
    _check(args, T, [[6, 31], [1, 2, 40], [64]][k % 3], seed=100 + k)


TN2_CASES = [(32, 8, 1, 4, [9, 20]), (64, 16, 2, 8, [33, 5, 1]), (96, 20, 1, 16, [40]), (128, 32, 2, 3, [64, 12])]


@pytest.mark.parametrize("case", TN2_CASES, ids=lambda c: f"F{c[0]}-K{c[1]}-L{c[2]}-q{c[3]}")
def test_tensornet2_shapes(hip_lib, case):
    from oracle import tn2_torch as T

    F, K, L, qd, sizes = case
    args = dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", embedding_dimension=F, num_rbf=K,
                num_layers=L, q_dim=qd, q_weights=[0.5 + 0.25 * i for i in range(L + 1)], cutoff_upper=4.5, max_z=12)
    q = torch.tensor([float(i % 3 - 1) for i in range(len(sizes))])
    _check(args, T, sizes, seed=7 * F + qd, q=q)


def test_tensornet2_nine_layers_through_the_tables(hip_lib):
    """TensorNet2 allows num_layers >= 9 when q_dim is small ((L + 1) q_dim <= 64); its L per-layer pair-block
    tables then exceed the eight slots of one interpolation launch -> they go in chunks of four.  Tables forced on for a
    short pair list (both the bucketed and the one-launch interpolation), and 8 + 1 TensorNet tables likewise."""
    from oracle import tn2_torch as T2
    from oracle import tensornet_torch as T

    args = dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", embedding_dimension=32, num_rbf=8,
                num_layers=9, q_dim=4, q_weights=[1.0] * 10, cutoff_upper=4.5, max_z=12)
    q = torch.tensor([1.0, -1.0])
    _check(args, T2, [20, 9], seed=77, q=q, options={"edge_table_min_pairs": 0})
    args = dict(W.TINY_ARGS, embedding_dimension=32, num_rbf=8, num_layers=7, cutoff_upper=4.5, max_z=12)
    _check(args, T, [20, 9], seed=78, options={"edge_table_min_pairs": 0})


def test_interleaved_models_sizes_and_streams(hip_lib):
    """Three engines alive at once, called in random order with batch sizes between 1 atom and ~3000 atoms (workspaces grow
    and are reused), on the default and on a side stream: every (model, input) pair returns bit-identical results each time
    it comes round again - nothing leaks between calls through a reused workspace, a cached count or the parameter block."""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(5)
    models = [create_model(dict(W.TINY_ARGS)).to("cuda"), create_model(dict(W.ET_TINY_ARGS)).to("cuda"),
              create_model(dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=4,
                                q_weights=[1.0, 1.0, 1.0])).to("cuda")]
    inputs = []
    for k, sizes in enumerate([[1], [64] * 40, [3, 17], [30] * 100, [90, 5, 64], [64] * 8, [2]]):
        z, pos, batch = _batch(sizes, 4000 + 50 * k, 12)
        inputs.append((z.cuda(), pos.cuda(), batch.cuda(), torch.zeros(len(sizes)).cuda()))
    side = torch.cuda.Stream()
    seen = {}
    gen = torch.Generator().manual_seed(9)
    for it in range(60):
        mi = int(torch.randint(0, len(models), (1,), generator=gen))
        ii = int(torch.randint(0, len(inputs), (1,), generator=gen))
        z, pos, batch, q = inputs[ii]
        use_side = it % 3 == 2
        if use_side:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side if use_side else torch.cuda.current_stream()):
            E, F = models[mi](z, pos, batch, q=q if mi != 1 else None)
        if use_side:
            torch.cuda.current_stream().wait_stream(side)
        E, F = E.detach().clone(), F.detach().clone()
        key = (mi, ii)
        if key in seen:
            assert torch.equal(E, seen[key][0]) and torch.equal(F, seen[key][1]), (it, key)
        else:
            assert torch.isfinite(E).all() and torch.isfinite(F).all()
            seen[key] = (E, F)
