"""-m gpu: branches of the HIP path that ship but were not evaluated on the GPU in round 1 ("What's weak" 1-2):
standardisation + Atomref in the head kernels, Ensemble.forward, a lower cutoff inside the model, rotational
equivariance (reference tests/test_equivariance.py), cell-list grids off the easy case (anisotropic, two cells per axis,
atoms on cell faces, sparse box, a box that changes between calls), cell list + static shapes + HIP-graph replay (the MD
mode of BASELINE configs[4]), input validation (reference: nn.Embedding / scatter raise), stale-graph detection, and the
RCCL path of the molecule-sharded evaluator at world size 1.  Oracles: oracle/tensornet_torch.py, oracle/et_torch.py
(both pinned to the unmodified reference in tests/test_oracle.py), oracle/tensornet_c.c; tolerance 1e-4 relative (fp32)."""
import os

import numpy as np
import pytest
import torch

from torchmdnet_amd import workloads as W

pytestmark = pytest.mark.gpu
REL = 1e-4

ROT = torch.tensor([[0.9886788, -0.1102370, 0.1017945],
                    [0.1363630, 0.9431761, -0.3030248],
                    [-0.0626055, 0.3134752, 0.9475304]])  # reference tests/test_equivariance.py:12-18


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _oracle(arch):
    if arch == "tensornet":
        from oracle import tensornet_torch as T
        return T
    from oracle import et_torch as T
    return T


def _args(arch, **kw):
    return dict(W.TINY_ARGS if arch == "tensornet" else W.ET_TINY_ARGS, **kw)


def _ragged(sizes, seed=700, z_max=19):
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(seed + m, n_atoms=n)
        zs.append(torch.from_numpy(zz) % z_max + 1)
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    return torch.cat(zs), torch.cat(ps), torch.cat(bs)


# ------------------------------------------------------------------ (a) mean / std / Atomref in the head kernels
@pytest.mark.parametrize("arch", ["tensornet", "equivariant-transformer"])
@pytest.mark.parametrize("sizes", [[7, 21, 1, 12], [300]])  # [300]: one molecule > 256 atoms -> k_head_energy + k_mol_sum
def test_mean_std_atomref(hip_lib, arch, sizes):
    from torchmdnet_amd.models.model import create_model

    T = _oracle(arch)
    args = _args(arch, prior_model="Atomref", prior_args={"max_z": 20})
    torch.manual_seed(21)
    model = create_model(dict(args), mean=torch.tensor(1.75), std=torch.tensor(0.375))
    table = torch.randn(20, 1) * 3.0
    with torch.no_grad():
        model.prior_model[0].atomref.weight.copy_(table)
    model = model.to("cuda")
    z, pos, batch = _ragged(sizes)
    if sizes == [300]:
        z, pos, batch = W.synthetic_batch(n_mol=1, n_atoms=300, first_seed=3)
        z = z % 19 + 1
    E, F = model(z.cuda(), pos.cuda(), batch.cuda())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert float(sd["mean"]) == 1.75 and float(sd["std"]) == 0.375
    Er, Fr = T.energy_and_forces(sd, T.hparams_from_args(args), z, pos, batch, atomref=table)
    assert rel_err(E.cpu(), Er) < REL and rel_err(F.cpu(), Fr) < REL
    # the prior and the standardisation really contribute: switching them off changes the numbers
    E0, F0 = T.energy_and_forces(dict(sd, mean=torch.tensor(0.0), std=torch.tensor(1.0)), T.hparams_from_args(args), z, pos, batch)
    assert rel_err(E.cpu(), E0) > 1e-2 and rel_err(F.cpu(), F0) > 1e-2
    # Atomref toggle (load_model(remove_ref_energy=False) flips it, reference model.py:250-259): the engine follows it
    model.prior_model[0].enable = False
    E1, F1 = model(z.cuda(), pos.cuda(), batch.cuda())
    Er1, Fr1 = T.energy_and_forces(sd, T.hparams_from_args(args), z, pos, batch)
    assert rel_err(E1.cpu(), Er1) < REL and rel_err(F1.cpu(), Fr1) < REL


# ------------------------------------------------------------------ (b) Ensemble.forward
@pytest.mark.parametrize("return_std", [False, True])
def test_ensemble_forward(hip_lib, return_std):
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import Ensemble, create_model

    models, refs = [], []
    z, pos, batch = _ragged([9, 14, 5])
    for s in (1, 2, 3):
        torch.manual_seed(s)
        m = create_model(dict(W.TINY_ARGS), mean=torch.tensor(0.1 * s), std=torch.tensor(1.0 + 0.2 * s))
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        refs.append(T.energy_and_forces(sd, T.hparams_from_args(W.TINY_ARGS), z, pos, batch))
        models.append(m.to("cuda"))
    ens = Ensemble(models, return_std=return_std)
    out = ens(z.cuda(), pos.cuda(), batch.cuda())
    ys, fs = torch.stack([r[0] for r in refs]), torch.stack([r[1] for r in refs])
    assert len(out) == (4 if return_std else 2)
    assert rel_err(out[0].cpu(), ys.mean(0)) < REL and rel_err(out[1].cpu(), fs.mean(0)) < REL
    if return_std:
        assert rel_err(out[2].cpu(), ys.std(0)) < 1e-3 and rel_err(out[3].cpu(), fs.std(0)) < 1e-3


# ------------------------------------------------------------------ (c) lower cutoff inside the model
@pytest.mark.parametrize("arch", ["tensornet", "equivariant-transformer"])
def test_cutoff_lower_in_model(hip_lib, arch):
    from torchmdnet_amd.models.model import create_model

    T = _oracle(arch)
    args = _args(arch, cutoff_lower=1.2, cutoff_upper=4.5)
    torch.manual_seed(5)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = _ragged([18, 30, 4, 11], seed=90)  # min distance 0.9 A: some pairs fall below the lower cutoff
    E, F = model(z.cuda(), pos.cuda(), batch.cuda())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    Er, Fr = T.energy_and_forces(sd, T.hparams_from_args(args), z, pos, batch)
    assert rel_err(E.cpu(), Er) < REL and rel_err(F.cpu(), Fr) < REL
    # it is a different pair set than with cutoff_lower = 0 (the branch is exercised)
    Er0, _ = T.energy_and_forces(sd, T.hparams_from_args(dict(args, cutoff_lower=0.0)), z, pos, batch)
    assert rel_err(Er0, Er) > 1e-3


# ------------------------------------------------------------------ (d) rotational equivariance
@pytest.mark.parametrize("arch,group", [("tensornet", "O(3)"), ("tensornet", "SO(3)"), ("equivariant-transformer", None)])
def test_rotation_equivariance(hip_lib, arch, group):
    """reference tests/test_equivariance.py: scalars invariant, vectors (here: forces) rotate with the frame.  Also an
    improper rotation: the O(3) TensorNet and the ET are reflection-equivariant too."""
    from torchmdnet_amd.models.model import create_model

    args = _args(arch)
    if group:
        args["equivariance_invariance_group"] = group
    torch.manual_seed(1234)
    model = create_model(dict(args)).to("cuda")
    z = torch.ones(100, dtype=torch.long)
    pos = torch.randn(100, 3)
    batch = torch.arange(50, dtype=torch.long).repeat_interleave(2)
    z2, pos2, batch2 = _ragged([40, 25], seed=31)
    for zz, pp, bb in ((z, pos, batch), (z2, pos2, batch2)):
        E, F = model(zz.cuda(), pp.cuda(), bb.cuda())
        Er, Fr = model(zz.cuda(), (pp @ ROT).cuda(), bb.cuda())
        assert rel_err(Er, E) < 2e-5
        assert rel_err(Fr.cpu(), F.cpu() @ ROT) < 2e-5
        if group != "SO(3)":
            refl = ROT @ torch.diag(torch.tensor([1.0, 1.0, -1.0]))
            Em, Fm = model(zz.cuda(), (pp @ refl).cuda(), bb.cuda())
            assert rel_err(Em, E) < 2e-5 and rel_err(Fm.cpu(), F.cpu() @ refl) < 2e-5


# ------------------------------------------------------------------ (e) cell list off the easy case
def _periodic_system(n, box_diag, seed, min_dist=0.8, skew=None):
    rng = np.random.default_rng(seed)
    L = np.asarray(box_diag, dtype=np.float64)
    pts = np.empty((0, 3))
    while pts.shape[0] < n:
        cand = rng.uniform(0.0, 1.0, size=(4 * n, 3)) * L
        for c in cand:
            if pts.shape[0] == n:
                break
            d = pts - c
            d -= np.round(d / L) * L
            if pts.shape[0] == 0 or np.min(np.linalg.norm(d, axis=1)) >= min_dist:
                pts = np.vstack([pts, c])
    box = np.diag(L)
    if skew is not None:
        frac = pts / L
        box = box + np.asarray(skew, dtype=np.float64)
        pts = frac @ box
    z = rng.integers(1, 20, size=n)
    return torch.from_numpy(z), torch.from_numpy(pts.astype(np.float32)), torch.from_numpy(box.astype(np.float32))


def _brute_vs_cell(model, z, pos, box, expect_grid=None):
    n = z.shape[0]
    batch = torch.zeros(n, dtype=torch.long)
    model.cell_list_min_atoms = 10 ** 9
    Eb, Fb = model(z.cuda(), pos.cuda(), batch.cuda(), box=box.cuda())
    cb = model._engine.counts
    assert model.cell_grid(n)[3] == 0
    model.cell_list_min_atoms = 1
    Ec, Fc = model(z.cuda(), pos.cuda(), batch.cuda(), box=box.cuda())
    cc = model._engine.counts
    grid = model.cell_grid(n)
    assert grid[3] == 1, "the cell list did not run"
    if expect_grid is not None:
        assert grid[:3] == tuple(expect_grid), grid
    assert cb[:2] == cc[:2], (cb, cc)  # same pair / edge counts
    assert rel_err(Ec, Eb) < 1e-5 and rel_err(Fc, Fb) < 1e-5
    return Ec, Fc


def test_cell_list_anisotropic_faces_sparse_small(hip_lib):
    from oracle import tensornet_c as CO, tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    args = dict(W.TINY_ARGS, max_num_neighbors=128)
    model = create_model(dict(args)).to("cuda")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    hp = T.hparams_from_args(args)
    rc = args["cutoff_upper"]
    # anisotropic orthorhombic box: 3 x 5 x 7 cells
    z, pos, box = _periodic_system(1400, (3.1 * rc, 5.1 * rc, 7.1 * rc), seed=1)
    Ec, Fc = _brute_vs_cell(model, z, pos, box, expect_grid=(3, 5, 7))
    Er, Fr = CO.energy_forces(sd, hp, z, pos, torch.zeros_like(z), box=box)
    assert rel_err(Ec.cpu(), Er) < REL and rel_err(Fc.cpu(), Fr) < REL
    # atoms exactly ON cell faces / box faces (and just outside the primary cell): binning must stay consistent
    z, pos, box = _periodic_system(900, (4.0 * rc, 4.0 * rc, 4.0 * rc), seed=2, min_dist=1.0)
    cw = rc  # cell width = L / 4 = rc exactly
    pos[:200] = torch.round(pos[:200] / cw) * cw  # lattice points of the cell grid (includes 0 and L)
    pos[:200] += torch.from_numpy(np.random.default_rng(3).integers(-1, 2, size=(200, 3)) * 4.0 * rc).float()  # whole-box shifts
    keep = torch.ones(900, dtype=torch.bool)  # drop coincident lattice points (d = 0 between distinct atoms)
    seen = set()
    for i in range(200):
        key = tuple(((pos[i] / cw).round().long() % 4).tolist())
        keep[i] = key not in seen
        seen.add(key)
    z, pos = z[keep], pos[keep]
    _brute_vs_cell(model, z, pos, box, expect_grid=(4, 4, 4))
    # sparse box: most cells empty (12^3 = 1728 cells, 150 atoms), grid capped by the cell_start capacity 8 N
    z, pos, box = _periodic_system(150, (12.2 * rc,) * 3, seed=4)
    pos = pos * 0.3 + torch.tensor([1.0, 50.0, -7.0])  # clustered in one corner, out of the primary cell
    Es, Fs = _brute_vs_cell(model, z, pos, box)
    assert max(model.cell_grid(150)[:3]) <= 10  # floor(cbrt(8 * 150)) = 10 < 12
    # two cells per axis (box = 2.4 rc): the 27 offsets hit each cell several times, every cell must be visited once
    z, pos, box = _periodic_system(260, (2.4 * rc, 2.6 * rc, 3.2 * rc), seed=5)
    _brute_vs_cell(model, z, pos, box, expect_grid=(2, 2, 3))
    # triclinic with an anisotropic grid (perpendicular widths, not edge lengths, set the grid)
    z, pos, box = _periodic_system(1100, (3.4 * rc, 4.6 * rc, 5.3 * rc), seed=6, skew=[[0, 0, 0], [1.6 * rc, 0, 0], [-1.1 * rc, 2.0 * rc, 0]])
    Et, Ft = _brute_vs_cell(model, z, pos, box)
    Er, Fr = CO.energy_forces(sd, hp, z, pos, torch.zeros_like(z), box=box)
    assert rel_err(Et.cpu(), Er) < REL and rel_err(Ft.cpu(), Fr) < REL


def test_cell_grid_follows_the_box_of_each_call(hip_lib):
    """a box that shrinks across a floor(w / rc) boundary while living at the same address (NPT / OpenMM rebuild
    the tensor every step; the caching allocator returns the same pointer) must get a new grid, not a cached one."""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    args = dict(W.TINY_ARGS, max_num_neighbors=160)
    model = create_model(dict(args)).to("cuda")
    rc = args["cutoff_upper"]
    z, pos, box = _periodic_system(1200, (5.05 * rc,) * 3, seed=11)
    box_dev = box.cuda()
    n = z.shape[0]
    batch = torch.zeros(n, dtype=torch.long).cuda()
    model.cell_list_min_atoms = 1
    E5, F5 = model(z.cuda(), pos.cuda(), batch, box=box_dev)
    assert model.cell_grid(n)[:3] == (5, 5, 5)
    scale = 4.9 / 5.05
    box_dev.mul_(scale)  # same tensor, same address, new values: 4 cells per axis now
    pos2 = (pos * scale).cuda()
    E4, F4 = model(z.cuda(), pos2, batch, box=box_dev)
    assert model.cell_grid(n)[:3] == (4, 4, 4)
    model.cell_list_min_atoms = 10 ** 9
    Eb, Fb = model(z.cuda(), pos2, batch, box=box_dev)
    assert rel_err(E4, Eb) < 1e-5 and rel_err(F4, Fb) < 1e-5
    # a fresh CPU / fp64 box every call (what ASE / OpenMM hand over) goes through the same device-side grid
    model.cell_list_min_atoms = 1
    E4b, F4b = model(z.cuda(), pos2, batch, box=(box * scale).double())
    assert rel_err(E4b, Eb) < 1e-5 and rel_err(F4b, Fb) < 1e-5


def test_cell_list_static_shapes_graph_replay(hip_lib):
    """BASELINE configs[4]'s MD mode: one periodic system, cell-list neighbours, static shapes, the whole E+F step replayed
    from a captured HIP graph over several position sets (and a changed box) - against the dynamic brute-force path."""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(0)
    args = dict(W.TINY_ARGS, max_num_neighbors=96)
    dyn = create_model(dict(args)).to("cuda")
    dyn.cell_list_min_atoms = 10 ** 9
    sta = create_model(dict(args, static_shapes=True)).to("cuda")
    sta.load_state_dict(dyn.state_dict())
    sta.cell_list_min_atoms = 1
    z, pos, box = W.water_box(n_side=8, spacing=3.1)  # 1536 atoms, 24.8 A box
    z = z % 19 + 1
    n = z.shape[0]
    z, pos, box = z.cuda(), pos.cuda(), box.cuda()
    batch = torch.zeros(n, dtype=torch.long, device="cuda")
    replay = sta.capture(z, pos, batch, box=box)
    assert sta.cell_grid(n)[3] == 1 and min(sta.cell_grid(n)[:3]) >= 3
    g = torch.Generator(device="cuda").manual_seed(5)
    for step in range(4):
        new = pos + 0.05 * step * torch.randn(pos.shape, generator=g, device="cuda")
        E, F = replay(new)
        E, F = E.clone(), F.clone()
        Er, Fr = dyn(z, new.clone(), batch, box=box)
        assert rel_err(E, Er) < 1e-5 and rel_err(F, Fr) < 1e-5, step
        sta.check_overflow(n, 1)
    # the box changes under the captured graph (barostat): the grid is recomputed on the device inside the replay
    with torch.no_grad():
        box.mul_(0.97)
    new = pos * 0.97
    E, F = replay(new)
    Er, Fr = dyn(z, new.clone(), batch, box=box)
    assert rel_err(E, Er) < 1e-5 and rel_err(F, Fr) < 1e-5


# ------------------------------------------------------------------ (f) input validation
@pytest.mark.parametrize("arch", ["tensornet", "equivariant-transformer"])
def test_input_validation(hip_lib, arch):
    from torchmdnet_amd.models.model import create_model

    args = _args(arch)
    model = create_model(dict(args)).to("cuda")
    z, pos, batch = _ragged([6, 9])
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    E, F = model(zc, pc, bc)
    zb = zc.clone()
    zb[3] = args["max_z"]  # nn.Embedding raises IndexError in the reference (tensornet.py:473, torchmd_et.py:144)
    with pytest.raises(IndexError, match="atomic number out of range"):
        model(zb, pc, bc)
    zb[3] = -1
    with pytest.raises(IndexError, match="atomic number out of range"):
        model(zb, pc, bc)
    bb = bc.clone()
    bb[-1] = 7
    with pytest.raises(RuntimeError, match="batch index out of range"):
        model(zc, pc, bb, num_systems=2)
    with pytest.raises(RuntimeError, match="same AMD GPU"):
        model(z, pc, bc)  # a host pointer must never reach a kernel
    with pytest.raises(RuntimeError, match="same AMD GPU"):
        model(zc, pc, batch)
    with pytest.raises(ValueError, match="one entry per atom"):
        model(zc[:-1], pc, bc)
    # the engine still works after the error returns
    E2, F2 = model(zc, pc, bc)
    assert torch.equal(E, E2) and torch.equal(F, F2)
    # static shapes: nothing is read back before the kernels run, so they run on clamped numbers and the poll raises
    sta = create_model(dict(args, static_shapes=True)).to("cuda")
    sta.load_state_dict(model.state_dict())
    Es, Fs = sta(zc, pc, bc)
    assert rel_err(Es, E) < 1e-5
    zb[3] = 10 ** 6
    with pytest.raises(IndexError, match="atomic number out of range"):
        sta(zb, pc, bc)
    with pytest.raises(RuntimeError, match="batch index out of range"):
        sta(zc, pc, bb, num_systems=2)
    Es2, _ = sta(zc, pc, bc)
    assert torch.equal(Es, Es2)


def test_stale_graph_replay_raises(hip_lib, golden_dir):
    """A captured graph holds raw pointers into the engine's parameter block and workspaces; once those are re-created
    (parameter change + eager call, or an eager call with a larger system) replaying it must raise, not touch freed memory."""
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    sta = create_model(dict(g["args"], static_shapes=True))
    sta.load_state_dict(g["state_dict"])
    sta = sta.to("cuda")
    z, pos, batch = g["z"].cuda(), g["pos"].cuda(), g["batch"].cuda()
    replay = sta.capture(z, pos, batch)
    E1, _ = replay()
    assert rel_err(E1.cpu(), g["E_q0"]) < REL
    zb, pb, bb = W.synthetic_batch(n_mol=40, n_atoms=60)
    sta(zb.cuda() % 19 + 1, pb.cuda(), bb.cuda())  # larger system: workspaces grow
    with pytest.raises(RuntimeError, match="stale HIP graph"):
        replay()
    replay = sta.capture(z, pos, batch)
    with torch.no_grad():
        sta.std.mul_(2.0)
    sta(z, pos, batch)  # parameter change + eager call: the handle is re-created
    with pytest.raises(RuntimeError, match="stale HIP graph"):
        replay()


# ------------------------------------------------------------------ (g) sharded evaluator over RCCL, world size 1
def test_sharded_evaluator_hip_model_nccl_world1(hip_lib):
    """The molecule-sharded evaluator with the HIP model as its compute callable and the `nccl` (= RCCL)
    backend: on the one-GPU box the world size is 1, which still loads RCCL, creates the communicator and runs the
    all-reduce of the energies on the device."""
    import socket

    import torch.distributed as dist
    from torchmdnet_amd.models.model import create_model
    from torchmdnet_amd.parallel import ShardedEvaluator

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        model = create_model(dict(W.TINY_ARGS)).to("cuda")
        z, pos, batch = _ragged([9, 14, 5, 20, 11], seed=40)
        q = torch.tensor([0.0, 1.0, -1.0, 0.0, 2.0])
        Er, Fr = model(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())

        def compute(zl, pl, bl, boxl, ql, nm):
            return model.energy_and_forces(zl, pl, bl, boxl, ql, nm, want_forces=True)

        for gather in (False, True):
            ev = ShardedEvaluator(compute, gather_forces=gather)
            E, F, (a0, a1) = ev.evaluate(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())
            t = torch.ones(4, device="cuda")
            dist.all_reduce(t)  # the RCCL communicator exists and runs on this device
            assert torch.equal(t, torch.ones(4, device="cuda"))
            assert (a0, a1) == (0, z.shape[0])
            assert torch.equal(E.view(-1, 1), Er) and torch.equal(F, Fr)
        # stepping form: shard cut once, local evaluation captured into one HIP graph (static shapes), all-reduce per step
        sta = create_model(dict(W.TINY_ARGS, static_shapes=True))
        sta.load_state_dict(model.state_dict())
        sta = sta.to("cuda")

        def compute_static(zl, pl, bl, boxl, ql, nm):
            return sta.energy_and_forces(zl, pl, bl, boxl, ql, nm, want_forces=True)

        ev = ShardedEvaluator(compute_static)
        ses = ev.prepare(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda(), graph=True,
                         guard=lambda: (id(sta._engine), sta._engine.generation))
        assert ses.graph is not None
        for k in range(3):
            newpos = (pos + 0.02 * k * torch.randn(pos.shape, generator=torch.Generator().manual_seed(k))).cuda()
            Es, Fs, (a0, a1) = ses.step(newpos)
            Ek, Fk = model(z.cuda(), newpos.clone(), batch.cuda(), q=q.cuda())
            assert rel_err(Es.view(-1, 1), Ek) < 1e-5 and rel_err(Fs, Fk) < 1e-5, k
        with torch.no_grad():
            sta.mean.add_(1.0)
        sta(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())  # re-uploads the parameters: the captured pointers are stale now
        with pytest.raises(RuntimeError, match="stale HIP graph"):
            ses.step()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ (h) the cell strategy of the neighbour operator
@pytest.mark.parametrize("box_kind", [None, "rectangular", "triclinic"])
@pytest.mark.parametrize("n_mol", [1, 5])
def test_neighbor_operator_cell_strategy_large(hip_lib, box_kind, n_mol):
    """OptimizedDistance(strategy="cell") runs tn_cell.hip (fictitious bounding box without PBC, batch-aware, triclinic
    too - the reference's cell kernel is orthorhombic-only) and returns the numpy specification's pair set."""
    from oracle.neighbors_numpy import reference_neighbors, sort_neighbors
    from torchmdnet_amd.models.utils import OptimizedDistance

    torch.manual_seed(7)
    n = 2600
    lbox = 26.0
    pos = torch.rand(n, 3) * lbox - 40.0
    batch = torch.sort(torch.randint(0, n_mol, (n,)))[0]
    box = None
    if box_kind == "rectangular":
        box = torch.tensor([[lbox, 0, 0], [0, lbox * 1.3, 0], [0, 0, lbox * 0.9]])
    elif box_kind == "triclinic":
        box = torch.tensor([[lbox, 0, 0], [4.0, lbox, 0], [-3.0, 5.0, lbox]])
    for loop, tr in ((True, True), (False, False)):
        ref = reference_neighbors(pos.numpy(), batch.numpy(), loop, tr, 3.0, None if box is None else box.numpy())
        outs = []
        for strategy in ("cell", "brute"):
            nl = OptimizedDistance(cutoff_upper=3.0, max_num_pairs=-40, return_vecs=True, loop=loop, include_transpose=tr,
                                   strategy=strategy, box=box).to("cuda")
            nb, d, vec = nl(pos.cuda(), batch.cuda())
            outs.append(sort_neighbors(nb.cpu().numpy(), vec.cpu().numpy(), d.cpu().numpy()))
        for got in outs:
            assert np.array_equal(got[0], ref[0])
            assert np.allclose(got[1], ref[1], atol=1e-5) and np.allclose(got[2], ref[2], atol=1e-5)
        if not tr:
            assert (outs[0][0][0] > outs[0][0][1]).all()  # i > j orientation survives the cell-order renumbering


# ------------------------------------------------------------------ (i) torch custom ops: opcheck, compile, export
def test_custom_ops_opcheck_compile_export(hip_lib, golden_dir):
    """The model's forward is one registered torch op (tmdnet::energy_forces; fake + autograd registered), so the tracing
    front ends of the reference's deployment paths work: torch.library.opcheck, torch.compile (calculators.py:297) and
    torch.export (tests/test_export.py)."""
    from torchmdnet_amd import ops
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    model = create_model(dict(g["args"], derivative=False))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda").eval()
    z, pos, batch = g["z"].cuda(), g["pos"].cuda(), g["batch"].cuda()
    n_mol = int(batch.max()) + 1
    y0, _ = model(z, pos, batch)
    key = model._engine.op_key
    pg = pos.clone().requires_grad_(True)
    torch.library.opcheck(torch.ops.tmdnet.energy_forces, (z, pg, batch, None, None, key, n_mol, True),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.tmdnet.neighbor_pairs, (pg, batch, None, 0.0, 4.0, 600, True, True, 0, n_mol),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    # autograd through the op == the forces
    y, _ = model(z, pg, batch)
    y.sum().backward()
    assert rel_err(-pg.grad.cpu(), g["F_q0"]) < REL
    # torch.compile: dynamo traces the module, the op stays one node (aot_eager: no code generation involved)
    compiled = torch.compile(model, backend="aot_eager")
    with torch.no_grad():
        yc, _ = compiled(z, pos, batch, num_systems=n_mol)
    assert torch.equal(yc, y0)
    pc = pos.clone().requires_grad_(True)
    yc2, _ = compiled(z, pc, batch, num_systems=n_mol)
    yc2.sum().backward()
    assert rel_err(-pc.grad.cpu(), g["F_q0"]) < REL
    # torch.export: the exported program contains the op and reproduces the energies
    ep = torch.export.export(model, (z, pos, batch), kwargs={"num_systems": n_mol}, strict=False)
    assert "tmdnet.energy_forces" in str(ep.graph)
    ye, _ = ep.module()(z, pos, batch, num_systems=n_mol)
    assert torch.equal(ye, y0)


# ------------------------------------------------------------------ (j) radial tables of the per-pair functions
def test_radial_tables_follow_parameter_edits(hip_lib, golden_dir):
    """The tables are a function of the radial weights: an upload marks them stale, the next call (or a question about them)
    rebuilds them - several edits between two calls cost one rebuild, and the results track the new weights exactly as the
    direct evaluation does."""
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    z, pos, batch, q = g["z"].cuda(), g["pos"].cuda(), g["batch"].cuda(), g["q"].cuda()
    model.set_engine_option("edge_table_min_pairs", 0)
    E0, F0 = model(z, pos, batch, q=q)
    assert rel_err(E0.cpu(), g["E"]) < REL and rel_err(F0.cpu(), g["F"]) < REL
    with torch.no_grad():  # two edits of radial weights, one of a non-radial one
        for name, p_ in model.named_parameters():
            if name.endswith("distance_proj1.weight") or name.endswith("linears_scalar.0.weight"):
                p_.mul_(1.07)
        model.representation_model.out_norm.bias.add_(0.01)
    E1, F1 = model(z, pos, batch, q=q)  # re-upload, tables rebuilt by this call
    assert model.engine_info("edge_table_T") >= 8192 and model.engine_info("edge_table_err_value") < 1e-6
    assert rel_err(E1, E0) > 1e-4  # the weights did change the answer
    model.set_engine_option("edge_table_min_pairs", 10 ** 12)  # same weights through the value + tangent GEMMs
    E2, F2 = model(z, pos, batch, q=q)
    assert rel_err(E1, E2) < 2e-6 and rel_err(F1, F2) < 2e-6


def test_edge_tables_equal_direct_evaluation(hip_lib, golden_dir):
    """Q(d) and every layer's edge MLP w(d) are functions of the distance alone: tabulated at parameter upload (verified
    there against the direct evaluation at all interval midpoints) and Hermite-interpolated per pair.  The table path must
    reproduce the direct pair-row GEMMs to fp32 rounding, on fixtures of the unmodified reference, with a lower cutoff,
    with total charges, under static shapes + graph replay, and at the full C2 size."""
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    z, pos, batch, q = g["z"].cuda(), g["pos"].cuda(), g["batch"].cuda(), g["q"].cuda()
    assert model.engine_info("edge_table_T") >= 8192  # built and verified at upload
    assert model.engine_info("edge_table_err_value") < 1e-6 and model.engine_info("edge_table_err_slope") < 2e-5
    model.set_engine_option("edge_table_min_pairs", 10 ** 12)  # the value + tangent GEMMs on the pair rows
    Ed, Fd = model(z, pos, batch, q=q)
    model.set_engine_option("edge_table_min_pairs", 0)  # the tables (a pair list this small: the one-launch interpolation)
    Et, Ft = model(z, pos, batch, q=q)
    assert rel_err(Et, Ed) < 2e-6 and rel_err(Ft, Fd) < 2e-6
    assert rel_err(Et.cpu(), g["E"]) < REL and rel_err(Ft.cpu(), g["F"]) < REL
    assert rel_err(Et.cpu().double(), g["E64"]) < REL and rel_err(Ft.cpu().double(), g["F64"]) < REL
    # energies only (no tangents are written)
    model.derivative = False
    with torch.no_grad():
        y, _ = model(z, pos, batch, q=q)
    assert rel_err(y.view(-1), Ed.view(-1)) < 2e-6
    model.derivative = True
    # lower cutoff (the self pair sits below the grid: it has its own exact row), sharper basis (K = 64)
    for extra in (dict(cutoff_lower=1.2, cutoff_upper=4.5), dict(num_rbf=64)):
        args = dict(W.TINY_ARGS, **extra)
        torch.manual_seed(8)
        m2 = create_model(dict(args)).to("cuda")
        z2, p2, b2 = _ragged([18, 30, 4, 11], seed=90)
        E0, F0 = m2(z2.cuda(), p2.cuda(), b2.cuda())
        m2.set_engine_option("edge_table_min_pairs", 0)
        assert m2.engine_info("edge_table_T") >= 8192
        E1, F1 = m2(z2.cuda(), p2.cuda(), b2.cuda())
        assert rel_err(E1, E0) < 2e-6 and rel_err(F1, F0) < 2e-6, extra
        sd = {k: v.detach().cpu() for k, v in m2.state_dict().items()}
        Er, Fr = T.energy_and_forces(sd, T.hparams_from_args(args), z2, p2, b2)
        assert rel_err(E1.cpu(), Er) < REL and rel_err(F1.cpu(), Fr) < REL, extra
    # static shapes: the sort runs over the pair capacity, replayed from a HIP graph
    sta = create_model(dict(g["args"], static_shapes=True))
    sta.load_state_dict(g["state_dict"])
    sta = sta.to("cuda")
    sta.set_engine_option("edge_table_min_pairs", 0)
    replay = sta.capture(z, pos, batch, q=q)
    for step in range(3):
        new = pos + 0.04 * step * torch.randn(pos.shape, generator=torch.Generator().manual_seed(step)).cuda()
        Es, Fs = replay(new)
        Er, Fr = model(z, new.clone(), batch, q=q)
        assert rel_err(Es, Er) < 2e-6 and rel_err(Fs, Fr) < 2e-6, step
    # full C2 batch: tables (default at this size) vs direct GEMMs
    torch.manual_seed(0)
    big = create_model(dict(W.C2_ARGS)).to("cuda")
    zb, pb, bb = (t.cuda() for t in W.synthetic_batch(n_mol=256))
    Et, Ft = big(zb, pb, bb)
    big.set_engine_option("edge_table_min_pairs", 10 ** 12)
    Ed, Fd = big(zb, pb, bb)
    assert rel_err(Et, Ed) < 2e-6 and rel_err(Ft, Fd) < 2e-6


# ------------------------------------------------------------------ single-system latency path (round 2)
@pytest.mark.parametrize("sizes", [[1], [5], [64], [17, 40, 3, 60], [200], [256], [257], [100, 180]])
@pytest.mark.parametrize("periodic", [False, True])
def test_small_system_graph_kernel_equals_general_path(hip_lib, sizes, periodic):
    """Static shapes build the neighbour graph of <= 256 atoms in ONE single-block launch (k_graph_small); the eager call
    uses the general seven-kernel path: same pairs in the same order -> energies and forces agree to rounding of the
    different sweep kernels; 257 / 280 atoms take the general path in both.  Unsorted molecule indices too."""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(11)
    z, pos, batch = _ragged(sizes, seed=910)
    box = None
    if periodic:
        if len(sizes) > 1:
            pytest.skip("one box per call in this test")
        box = torch.tensor([[13.0, 0.0, 0.0], [1.5, 12.0, 0.0], [-1.0, 2.0, 14.0]])
    dyn = create_model(_args("tensornet")).to("cuda")
    sta = create_model(_args("tensornet", static_shapes=True, max_num_neighbors=64))
    sta.load_state_dict(dyn.state_dict())
    sta = sta.to("cuda")
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    bx = None if box is None else box.cuda()
    E0, F0 = dyn(zc, pc, bc, box=bx)
    E1, F1 = sta(zc, pc, bc, box=bx)
    assert dyn._engine.counts[:2] == sta._engine.counts[:2]
    assert rel_err(E1, E0) < 2e-6 and rel_err(F1, F0) < 2e-6
    if len(sizes) > 1:  # unsorted batch: the graph kernels scan all atoms for the members of a molecule
        perm = torch.randperm(z.shape[0], generator=torch.Generator().manual_seed(3)).cuda()
        E2, F2 = sta(zc[perm], pc[perm], bc[perm])
        assert rel_err(E2, E0) < 2e-6 and rel_err(F2, F0[perm]) < 2e-6
    # the HIP graph of the captured step replays it
    replay = sta.capture(zc, pc, bc, box=bx)
    moved = pc + 0.03 * torch.randn(pc.shape, generator=torch.Generator().manual_seed(5)).cuda()
    E3, F3 = replay(moved)
    E4, F4 = dyn(zc, moved.clone(), bc, box=bx)
    assert rel_err(E3, E4) < 2e-6 and rel_err(F3, F4) < 2e-6
    assert replay.pos.data_ptr() != moved.data_ptr() and torch.equal(replay.pos, moved)


def test_one_launch_interpolation_equals_bucketed(hip_lib, golden_dir, tmp_path):
    """Short pair lists interpolate the radial tables in one launch (k_edge_interp_direct); TMDNET_EDGE_DIRECT_MAX=0 in a
    fresh process forces the bucketed batch path on the same input: same arithmetic on the same distance bits (the compiler
    contracts the two kernels' multiply-adds differently: agreement to the last bits, not bit for bit)."""
    import subprocess
    import sys

    from torchmdnet_amd.models.model import create_model

    g = torch.load(os.path.join(golden_dir, "tiny_ref.pt"))
    model = create_model(dict(g["args"]))
    model.load_state_dict(g["state_dict"])
    model = model.to("cuda")
    E, F = model(g["z"].cuda(), g["pos"].cuda(), g["batch"].cuda(), q=g["q"].cuda())
    out = tmp_path / "bucketed.pt"
    code = (
        "import os, sys, torch\n"
        f"sys.path[:0] = {[p for p in sys.path if p]!r}\n"
        "from torchmdnet_amd.models.model import create_model\n"
        f"g = torch.load({os.path.join(golden_dir, 'tiny_ref.pt')!r})\n"
        "m = create_model(dict(g['args'])); m.load_state_dict(g['state_dict']); m = m.to('cuda')\n"
        "E, F = m(g['z'].cuda(), g['pos'].cuda(), g['batch'].cuda(), q=g['q'].cuda())\n"
        f"torch.save((E.cpu(), F.cpu()), {str(out)!r})\n"
    )
    env = dict(os.environ, TMDNET_EDGE_DIRECT_MAX="0")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
    Eb, Fb = torch.load(out)
    assert rel_err(E.detach().cpu(), Eb.detach()) < 1e-6 and rel_err(F.detach().cpu(), Fb.detach()) < 1e-6
