"""-m gpu: parity AT THE SIZES bench.py times (VERDICT r02 "parity holes").  The kernels that dominate a bench leg are chosen by
problem size and width (F % 64, pair count, molecule size), so a parity test on a small configuration does not exercise them:

* `water10k` routing: one periodic system, F = 128, cell list, `g.small_mols == 0` -> batched row sweep forward
  (`launch_message_v4` / `k_message`) and `k_message_adjoint_gd` reverse, radial tables, split-bf16 tensor linears.
* `tensornet2` leg: F = 128, q_dim = 16, 256 x 64 atoms -> the E-row split-bf16 GEMMs (403 802 rows) and the edge kernels.
* `et_c4` leg: ET-SPICE hyper-parameters on 256 x 64 atoms.

Reference behaviour matched: cell list == brute force (tests/test_neighbors.py:74-148), static shapes == dynamic
(tests/test_staticshapes.py:59-87).  Tolerance: 1e-4 relative (BASELINE north_star), measured as max|delta| / max|reference|."""
import pytest
import torch

from torchmdnet_amd import workloads as W

pytestmark = pytest.mark.gpu
REL = 1e-4


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("n_side", [10])
def test_water_box_c2_width_cell_and_brute_vs_oracle(hip_lib, n_side):
    """3000-atom periodic water box with the C2 model (F = 128, L = 2, K = 32): every atom's energy contribution and force
    against the scalar-C oracle (which handles boxes), through the cell list AND the brute-force sweep."""
    from oracle import tensornet_c as CO, tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    args = dict(W.C2_ARGS)
    model = create_model(dict(args)).to("cuda")
    z, pos, box = W.water_box(n_side=n_side, spacing=3.1)  # 3000 atoms, 31 A box -> 6 cells per axis
    pos = pos + torch.tensor([-7.0, 40.0, 3.0])  # atoms outside the primary cell: wrapping must not matter
    batch = torch.zeros_like(z)
    zc, pc, bc, xc = z.cuda(), pos.cuda(), batch.cuda(), box.cuda()
    model.cell_list_min_atoms = 10 ** 9
    Eb, Fb = model(zc, pc, bc, box=xc)
    counts_brute = model._engine.counts[:2]
    model.cell_list_min_atoms = 1
    Ec, Fc = model(zc, pc, bc, box=xc)
    assert model.cell_grid(z.shape[0])[3] == 1, "the cell list really ran"
    assert model._engine.counts[:2] == counts_brute
    assert model.engine_info("edge_table_T") >= 8192, "the radial tables are on (bench routing)"
    Ec2, Fc2 = model(zc, pc, bc, box=xc)
    assert torch.equal(Ec, Ec2) and torch.equal(Fc, Fc2)  # deterministic
    # cell order only renumbers the atoms: the sums run in another order, nothing else
    assert rel_err(Ec, Eb) < 1e-5 and rel_err(Fc, Fb) < 2e-5
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    Er, Fr = CO.energy_forces(sd, T.hparams_from_args(args), z, pos, batch, box=box)
    for E_, F_ in ((Ec, Fc), (Eb, Fb)):
        assert rel_err(E_.cpu(), Er) < REL
        assert rel_err(F_.cpu(), Fr) < REL
    # static shapes (capacity-sized launches, pair count on the device) through the same routing
    sta = create_model(dict(args, static_shapes=True))
    sta.load_state_dict(model.state_dict())
    sta = sta.to("cuda")
    sta.cell_list_min_atoms = 1
    Es, Fs = sta(zc, pc, bc, box=xc)
    assert rel_err(Es, Ec) < 1e-6 and rel_err(Fs, Fc) < 1e-5


def test_tensornet2_bench_scale_vs_oracle(hip_lib):
    """bench.py's `tensornet2` leg (F = 128, L = 2, q_dim = 16, S-mol64 256 x 64 atoms, all-to-all Coulomb): oracle
    `tn2_torch` on 3 sampled molecules, bit-identical repeat, radial tables on and off."""
    from oracle import tn2_torch as T2
    from torchmdnet_amd.models.model import create_model

    args = dict(W.C2_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=16, q_weights=[1.0, 1.0, 1.0])
    torch.manual_seed(0)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = W.synthetic_batch(n_mol=256)
    q = torch.tensor([float(m % 3 - 1) for m in range(256)])
    zc, pc, bc, qc = z.cuda(), pos.cuda(), batch.cuda(), q.cuda()
    E, F = model(zc, pc, bc, q=qc)
    E2, F2 = model(zc, pc, bc, q=qc)
    assert torch.equal(E, E2) and torch.equal(F, F2)
    assert torch.isfinite(E).all() and torch.isfinite(F).all()
    assert model.engine_info("edge_table_T") >= 8192
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    hp = T2.hparams_from_args(args)
    for m in (0, 129, 255):
        sel = batch == m
        Er, Fr = T2.energy_and_forces(sd, hp, z[sel], pos[sel], torch.zeros(int(sel.sum()), dtype=torch.long), q=q[m:m + 1])
        assert rel_err(E[m].cpu().reshape(1, 1), Er.reshape(1, 1)) < REL, m
        assert rel_err(F[sel.cuda()].cpu(), Fr) < REL, m
    net = torch.zeros(256, 3, device="cuda").index_add(0, bc, F)
    assert net.abs().max().item() < 1e-3 * F.abs().max().item()
    model.set_engine_option("edge_table_min_pairs", 10 ** 12)  # pair blocks through the value + tangent GEMMs
    Ed, Fd = model(zc, pc, bc, q=qc)
    assert rel_err(Ed, E) < 5e-6 and rel_err(Fd, F) < 2e-5


def test_et_c4_bench_scale_vs_oracle(hip_lib):
    """bench.py's `et_c4` leg (ET-SPICE hyper-parameters, 256 x 64 atoms, rc = 10 A -> 505 k pairs): oracle on 2 sampled molecules,
    bit-identical repeat, zero net force per molecule."""
    from oracle import et_torch as ET
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    model = create_model(dict(W.C4_ARGS)).to("cuda")
    z, pos, batch = W.synthetic_batch(n_mol=256)
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    E, F = model(zc, pc, bc)
    E2, F2 = model(zc, pc.clone(), bc)
    assert torch.equal(E, E2) and torch.equal(F, F2)
    net = torch.zeros(256, 3, device="cuda").index_add(0, bc, F)
    assert net.abs().max().item() < 1e-3 * F.abs().max().item()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    hp = ET.hparams_from_args(W.C4_ARGS)
    for m in (7, 200):
        sel = batch == m
        Eo, Fo = ET.energy_and_forces(sd, hp, z[sel], pos[sel], torch.zeros(int(sel.sum()), dtype=torch.long))
        assert rel_err(E[m].cpu().reshape(1, 1), Eo) < REL
        assert rel_err(F[sel.cuda()].cpu(), Fo) < REL


def test_nonperiodic_30k_atoms_through_the_cell_list(hip_lib):
    """One large NON-periodic system (a 32 k-atom water droplet cut from the lattice): the model takes the O(N) cell list with a
    fictitious box around the bounding box (reference models/utils.py:206-212: the cell strategy without a box), atoms
    renumbered in cell order by the hand-written counting sort (tn_cell.hip).  Same pair set and the same energies / forces as
    the brute-force sweep; forces of every atom against the scalar-C oracle."""
    from oracle import tensornet_c as CO, tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    args = dict(W.TINY_ARGS, max_num_neighbors=96)
    model = create_model(dict(args)).to("cuda")
    z, pos, _ = W.water_box(n_side=22, spacing=3.1)  # 31 944 atoms
    z = z % 19 + 1
    pos = pos + torch.tensor([-100.0, 37.0, 5.0])
    batch = torch.zeros_like(z)
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    Ec, Fc = model(zc, pc, bc)
    grid = model.cell_grid(z.shape[0])
    assert grid[3] == 1 and min(grid[:3]) >= 3, grid  # the cell list ran (default threshold: 1024 atoms)
    counts_cell = model._engine.counts[:2]
    Ec2, Fc2 = model(zc, pc.clone(), bc)
    assert torch.equal(Ec, Ec2) and torch.equal(Fc, Fc2)  # the placement atomics do not leak into the result
    model.cell_list_min_atoms = 10 ** 9
    Eb, Fb = model(zc, pc.clone(), bc)
    assert model.cell_grid(z.shape[0])[3] == 0 and model._engine.counts[:2] == counts_cell
    assert rel_err(Ec, Eb) < 1e-5 and rel_err(Fc, Fb) < 2e-5
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    Er, Fr = CO.energy_forces(sd, T.hparams_from_args(args), z, pos, batch)
    assert rel_err(Ec.cpu(), Er) < REL and rel_err(Fc.cpu(), Fr) < REL


@pytest.mark.parametrize("periodic", [True, False])
def test_several_large_molecules_through_the_cell_list(hip_lib, periodic):
    """Three 1536-atom systems that OVERLAP in space (same lattice, different jitter), in one common box or none: the model's
    cell list bins all atoms in one grid, atoms of different molecules are interleaved in cell order, pairs stay inside a
    molecule (reference: one `box` for the whole batch, models/utils.py:233-313).  Same numbers as the brute-force sweep inside
    each molecule, per-molecule energies / charges included; every molecule against the oracle."""
    from oracle import tensornet_c as CO, tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(4)
    args = dict(W.TINY_ARGS, max_num_neighbors=96)
    model = create_model(dict(args)).to("cuda")
    zs, ps, bs = [], [], []
    for m in range(3):
        z, pos, box = W.water_box(n_side=8, spacing=3.1, seed=10 + m)
        zs.append((z + m) % 19 + 1)
        ps.append(pos + 0.4 * m)
        bs.append(torch.full_like(z, m))
    z, pos, batch = torch.cat(zs), torch.cat(ps), torch.cat(bs)
    q = torch.tensor([1.0, 0.0, -2.0])
    kw = dict(box=box.cuda()) if periodic else {}
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    Ec, Fc = model(zc, pc, bc, q=q.cuda(), **kw)
    assert model.cell_grid(z.shape[0], 3)[3] == 1, "the cell list ran for several molecules"
    counts = model._engine.counts[:2]
    Ec2, Fc2 = model(zc, pc.clone(), bc, q=q.cuda(), **kw)
    assert torch.equal(Ec, Ec2) and torch.equal(Fc, Fc2)
    model.cell_list_min_atoms = 10 ** 9
    Eb, Fb = model(zc, pc.clone(), bc, q=q.cuda(), **kw)
    assert model.cell_grid(z.shape[0], 3)[3] == 0 and model._engine.counts[:2] == counts
    assert rel_err(Ec, Eb) < 1e-5 and rel_err(Fc, Fb) < 2e-5
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    hp = T.hparams_from_args(args)
    for m in range(3):
        sel = batch == m
        Er, Fr = CO.energy_forces(sd, hp, z[sel], pos[sel], torch.zeros(int(sel.sum()), dtype=torch.long), box=box if periodic else None,
                                  q=q[m:m + 1])
        assert rel_err(Ec[m].cpu().reshape(1, 1), Er) < REL and rel_err(Fc[sel.cuda()].cpu(), Fr) < REL, m
    # an out-of-range molecule index is reported like on the brute-force path
    model.cell_list_min_atoms = 1024
    bad = bc.clone()
    bad[5] = 7
    with pytest.raises(RuntimeError):
        model(zc, pc.clone(), bad, q=q.cuda(), num_systems=3, **kw)


def test_fused_tensor_linears_with_charges_ragged_partial_tile_vs_oracle(hip_lib):
    """The fused tensor linears (csrc/tn_tlin9.hip) at a size that engages them (>= 128 tiles of 32 atoms) on a batch that is NOT
    the benchmark's: ragged molecules (tile boundaries inside molecules, a partial last tile), total charges (the charge factor
    enters the update epilogue and the update-adjoint prologue, reference tensornet.py:789, 812), O(3) and SO(3) products.  Sampled
    molecules against the torch oracle (molecules are independent), all molecules against the unfused schedule of a second handle
    is not possible in one process (the switch is read once) - tests/test_gpu_switches.py runs the unfused schedule on fixtures."""
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    for group in ("O(3)", "SO(3)"):
        args = dict(W.C2_ARGS, equivariance_invariance_group=group)
        torch.manual_seed(5)
        model = create_model(dict(args)).to("cuda")
        sizes = [64 - (m % 7) for m in range(90)]  # 5 490 atoms = 172 tiles (one round), last tile partial
        zs, ps, bs = [], [], []
        for m, n in enumerate(sizes):
            zz, pp = W.synthetic_molecule(9000 + m, n_atoms=n)
            zs.append(torch.from_numpy(zz)); ps.append(torch.from_numpy(pp)); bs.append(torch.full((n,), m, dtype=torch.long))
        z, pos, batch = torch.cat(zs), torch.cat(ps), torch.cat(bs)
        assert z.shape[0] % 32 != 0
        q = torch.tensor([float(m % 3 - 1) for m in range(len(sizes))])
        E, F = model(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())
        E2, F2 = model(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())
        assert torch.equal(E, E2) and torch.equal(F, F2)  # deterministic
        sd = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
        hp = T.hparams_from_args(args)
        off = [0]
        for n in sizes:
            off.append(off[-1] + n)
        for m in (0, 37, 89):
            sl = slice(off[m], off[m + 1])
            Er, Fr = T.energy_and_forces(sd, hp, z[sl], pos[sl].double(), torch.zeros(sizes[m], dtype=torch.long), q=q[m:m + 1].double())
            assert abs(E[m].item() - Er.item()) / max(abs(Er.item()), 1e-12) < REL, (group, m)
            assert rel_err(F[sl].cpu().double(), Fr) < REL, (group, m)
        # zero net force per molecule (pairwise forces): every molecule, not only the sampled ones
        net = torch.zeros(len(sizes), 3, device="cuda").index_add_(0, batch.cuda(), F)
        assert net.abs().max().item() < 1e-3 * F.abs().max().item()


def test_recompute_pair_rows_is_bit_identical_and_drops_the_pair_workspace(hip_lib):
    """Option "recompute_pair_rows" (VERDICT r03 item 6): the sweeps of a large system interpolate the per-pair rows from the radial
    tables themselves; same arithmetic (csrc/tn_interp.h) -> energies and forces equal the stored-row path bit for bit, and the
    forward workspace shrinks by the per-pair rows (10 125-atom periodic water box, C2 model)."""
    import ctypes as C
    from torchmdnet_amd import _C, workloads as W
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    model = create_model(dict(W.C2_ARGS)).cuda()
    z, pos, box = (t.cuda() for t in W.water_box(n_side=15))
    batch = torch.zeros_like(z)
    E0, F0 = model.energy_and_forces(z, pos, batch, box, None, 1, True)
    e0 = model.energy_and_forces(z, pos, batch, box, None, 1, False)[0]
    n, (n_pairs, n_edges, _) = int(z.shape[0]), model._engine.counts
    lib, nb = _C.lib(), C.c_size_t(0)
    lib.tmdnet_forward_workspace_bytes(model._engine.handle, n, 1, n_pairs, n_edges, 1, C.byref(nb))
    stored = nb.value
    model.set_engine_option("recompute_pair_rows", 1)
    E1, F1 = model.energy_and_forces(z, pos, batch, box, None, 1, True)
    e1 = model.energy_and_forces(z, pos, batch, box, None, 1, False)[0]
    lib.tmdnet_forward_workspace_bytes(model._engine.handle, n, 1, n_pairs, n_edges, 1, C.byref(nb))
    assert torch.equal(E0, E1) and torch.equal(F0, F1) and torch.equal(e0, e1)
    assert nb.value < 0.45 * stored, (nb.value, stored)  # ~12.6 KB per pair gone; the per-atom tensors remain
    model.set_engine_option("recompute_pair_rows", 0)
    E2, F2 = model.energy_and_forces(z, pos, batch, box, None, 1, True)
    assert torch.equal(E0, E2) and torch.equal(F0, F2)


def test_quarter_million_atom_box_pair_set_and_determinism(hip_lib):
    """273 375-atom periodic water box: the cell list's pair set is the brute-force pair set, the step is reproducible bit for bit
    and translation-invariant.  This configuration holds a pair within one ulp of the cutoff; the count and the fill pass of the
    row once disagreed about it (their copies of |delta|^2 were contracted differently), the row got one entry too many, a real
    neighbour fell off its end and ~50 atoms had forces off by 10 % (round 4; csrc/tn_common.h pair_geometry is the fix)."""
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    z, pos, box = (t.cuda() for t in W.water_box(n_side=45))
    batch, n = torch.zeros_like(z), z.shape[0]
    keys = []
    for strategy in (0, 1):  # brute force, cell list
        nb, _, ds, npairs = torch.ops.tmdnet.neighbor_pairs(pos, batch, box, 0.0, 5.0, 9_000_000, False, False, strategy, 1)
        k = int(npairs.reshape(-1)[0])
        assert float(ds[:k].max()) <= 5.0  # |delta|^2 < 25 decides; its square root may round to 5
        keys.append(torch.sort(nb[0, :k] * n + nb[1, :k]).values)
    assert torch.equal(keys[0], keys[1])
    torch.manual_seed(0)
    model = create_model(dict(W.C2_ARGS, max_num_neighbors=96)).cuda()
    E, F = model(z, pos, batch, box=box)
    E1, F1 = model(z, pos, batch, box=box)
    assert torch.equal(E, E1) and torch.equal(F, F1)
    E2, F2 = model(z, pos + torch.tensor([1.234, -2.5, 7.7], device="cuda"), batch, box=box)
    assert abs(float(E2) - float(E)) < 1e-5 * abs(float(E))
    assert (F2 - F).abs().max().item() < 1e-3 * F.abs().max().item()


def test_thousands_of_tiny_molecules_through_the_batch_kernels(hip_lib):
    """3 000 molecules of 1 .. 9 atoms (C2 model): the batch-scale kernels on rows that are mostly one or two entries long - tiles whose
    column window spans a dozen molecules, balanced walks whose helper segments are empty, single atoms (self edge only).  A sample
    of molecules against the oracle, and every molecule's forces sum to zero."""
    import numpy as np
    from oracle import tensornet_torch as T
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    model = create_model(dict(W.C2_ARGS)).cuda()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    hp = T.hparams_from_args(W.C2_ARGS)
    rng = np.random.default_rng(11)
    sizes = rng.integers(1, 10, size=3000)
    pos = np.concatenate([rng.uniform(0, 1.6 * max(s, 2) ** (1 / 3), size=(s, 3)) for s in sizes]).astype(np.float32)
    z = torch.from_numpy(rng.choice([1, 6, 7, 8], size=int(sizes.sum())))
    batch = torch.from_numpy(np.repeat(np.arange(3000), sizes))
    pos = torch.from_numpy(pos)
    E, F = model(z.cuda(), pos.cuda(), batch.cuda())
    assert torch.isfinite(E).all() and torch.isfinite(F).all()
    net = torch.zeros(3000, 3, device="cuda").index_add_(0, batch.cuda(), F)
    assert net.abs().max().item() < 1e-4 * max(1.0, F.abs().max().item())
    for m in (0, 1, 2, 1499, 2999):
        sel = batch == m
        Eo, Fo = T.energy_and_forces(sd, hp, z[sel], pos[sel], torch.zeros(int(sel.sum()), dtype=torch.long))
        assert abs(E[m].item() - Eo.item()) < 1e-4 * max(1.0, abs(Eo.item())), m
        assert (F[sel.cuda()].cpu() - Fo).abs().max().item() < 1e-4 * max(1.0, Fo.abs().max().item()), m
