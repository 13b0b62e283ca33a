"""not-gpu: molecule sharding + energy reduction over torch.distributed (gloo, world_size 2).
The compute callable is the oracle (allowed in tests): the partition / collective logic is what is tested."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torchmdnet_amd.parallel import ShardedEvaluator, molecule_ranges


def test_molecule_ranges_cover_and_balance():
    sizes = [5, 64, 64, 3, 40, 17, 64, 9]
    batch = torch.cat([torch.full((n,), m, dtype=torch.long) for m, n in enumerate(sizes)])
    for world in (1, 2, 3, 4, 8, 11):
        r = molecule_ranges(batch, len(sizes), world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == len(sizes) and r[-1][3] == sum(sizes)
        for a, b in zip(r[:-1], r[1:]):
            assert a[1] == b[0] and a[3] == b[2]
        for (m0, m1, a0, a1) in r:
            assert a1 - a0 == sum(sizes[m0:m1])
    r = molecule_ranges(batch, len(sizes), 2)
    assert abs((r[0][3] - r[0][2]) - (r[1][3] - r[1][2])) <= 64


def _worker(rank, world, port, tmpdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "torchmd-net_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import tensornet_c as CO, tensornet_torch as T
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = create_model(dict(W.TINY_ARGS))
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    hp = T.hparams_from_args(W.TINY_ARGS)
    sizes = [9, 14, 5, 20, 11]
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(40 + m, n_atoms=n)
        zs.append(torch.from_numpy(zz)); ps.append(torch.from_numpy(pp)); bs.append(torch.full((n,), m, dtype=torch.long))
    z, pos, batch = torch.cat(zs), torch.cat(ps), torch.cat(bs)
    q = torch.tensor([0.0, 1.0, -1.0, 0.0, 2.0])

    def compute(zl, pl, bl, boxl, ql, nm):
        return CO.energy_forces(sd, hp, zl, pl, bl, q=ql)

    ev = ShardedEvaluator(compute, gather_forces=True)
    E, F, rng = ev.evaluate(z, pos, batch, q=q)
    Er, Fr = CO.energy_forces(sd, hp, z, pos, batch, q=q)
    ok = torch.allclose(E.view(-1, 1), Er, atol=1e-6) and torch.allclose(F, Fr, atol=1e-6) and rng == (0, z.shape[0])
    ev2 = ShardedEvaluator(compute, gather_forces=False)
    E2, Fl, (a0, a1) = ev2.evaluate(z, pos, batch, q=q)
    ok = ok and torch.allclose(E2.view(-1, 1), Er, atol=1e-6) and torch.allclose(Fl, Fr[a0:a1], atol=1e-6)
    # stepping form: shard cut once, static buffers, new positions per step (no graph on CPU)
    ses = ev2.prepare(z, pos, batch, q=q)
    for k in range(3):
        newpos = pos + 0.01 * k
        Es, Fs, (s0, s1) = ses.step(newpos)
        Ek, Fk = CO.energy_forces(sd, hp, z, newpos, batch, q=q)
        ok = ok and torch.allclose(Es.view(-1, 1), Ek, atol=1e-6) and torch.allclose(Fs, Fk[s0:s1], atol=1e-6) and (s0, s1) == (a0, a1)
    # weak form: every rank holds only its own molecules; energies land in the rank's slots of the global vector
    own = [m for m in range(len(sizes)) if m % world == rank]
    zl = torch.cat([zs[m] for m in own]); pl = torch.cat([ps[m] for m in own])
    bl = torch.cat([torch.full((sizes[m],), i, dtype=torch.long) for i, m in enumerate(own)])
    n_glob, lo = 3 * world, 3 * rank  # 3 slots per rank (rank 1 fills only 2 of them)
    sw = ShardedEvaluator(lambda zz, pp, bb, boxl, ql, nm: CO.energy_forces(sd, hp, zz, pp, bb)).prepare_local(zl, pl, bl, n_glob, lo)
    Ew, Fw, _ = sw.step()
    Eall, _ = CO.energy_forces(sd, hp, z, pos, batch)
    for r in range(world):
        for i, m in enumerate([m for m in range(len(sizes)) if m % world == r]):
            ok = ok and abs(float(Ew[3 * r + i]) - float(Eall[m])) < 1e-5
    ok = ok and (world < 2 or float(Ew[5]) == 0.0) and Fw.shape[0] == zl.shape[0]
    with open(os.path.join(tmpdir, f"ok{rank}"), "w") as fh:
        fh.write("1" if ok else "0")
    dist.destroy_process_group()


def test_sharded_evaluator_gloo_world2(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"ok{r}").read() == "1"


def _periodic_system(n, lengths, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 3, generator=g) * torch.tensor(lengths)
    z = torch.randint(1, 9, (n,), generator=g)
    return z, pos, torch.diag(torch.tensor(lengths))


def _spatial_setup():
    from oracle import tensornet_torch as T
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    args = dict(W.TINY_ARGS, num_layers=1, cutoff_upper=2.5, embedding_dimension=16, num_rbf=8)
    torch.manual_seed(3)
    model = create_model(dict(args))
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    sd["mean"] = torch.tensor(0.75)  # every rank's sum contains the per-molecule offset once: the evaluator must not add it twice
    hp = T.hparams_from_args(args)

    def compute(zl, pl, boxl, wl):
        return T.energy_and_forces(sd, hp, zl, pl, torch.zeros_like(zl), box=boxl, atom_weights=wl)

    return T, sd, hp, compute, args


def test_spatial_decomposition_single_process_equals_the_whole_system():
    """Slab decomposition with deep halo (parallel.SpatialEvaluator), the ranks evaluated one after the other: the sum of the ranks'
    energies and forces is the undecomposed periodic system's, for slabs wider and narrower than the halo (self images)."""
    from torchmdnet_amd.parallel import SpatialEvaluator

    T, sd, hp, compute, args = _spatial_setup()
    for lengths, n in (([17.0, 6.0, 5.5], 70), ([9.0, 5.5, 6.5], 45)):
        z, pos, box = _periodic_system(n, lengths, seed=len(lengths) + n)
        Er, Fr = T.energy_and_forces(sd, hp, z, pos, torch.zeros_like(z), box=box)
        for world in (2, 3):
            ev = SpatialEvaluator(compute, args["cutoff_upper"], args["num_layers"], energy_offset=0.75)
            E, F, n_local = torch.zeros(1), torch.zeros(n, 3), 0
            for r in range(world):
                e, f = ev.contribution(z, pos, box, r, world)
                E, F = E + e, F + f
                n_local += ev.local_system(pos, box, r, world)[0].numel()
            E = E + 0.75
            assert n_local > n  # there is a halo
            assert abs(float(E) - float(Er)) < 1e-5 * max(1.0, abs(float(Er))), (lengths, world)
            assert (F - Fr).abs().max().item() < 1e-5 * max(1.0, Fr.abs().max().item()), (lengths, world)


def _spatial_worker(rank, world, port, tmpdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "torchmd-net_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from torchmdnet_amd.parallel import SpatialEvaluator

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, sd, hp, compute, args = _spatial_setup()
    z, pos, box = _periodic_system(64, [16.0, 6.0, 5.5], seed=9)
    ev = SpatialEvaluator(compute, args["cutoff_upper"], args["num_layers"], energy_offset=0.75)
    E, F = ev.evaluate(z, pos, box)
    Er, Fr = T.energy_and_forces(sd, hp, z, pos, torch.zeros_like(z), box=box)
    ok = abs(float(E) - float(Er)) < 1e-5 * max(1.0, abs(float(Er))) and (F - Fr).abs().max().item() < 1e-5 * max(1.0, Fr.abs().max().item())
    ok = ok and 0 < ev.local_system(pos, box, rank, world)[3] < 64  # this rank owns a proper part of the atoms
    with open(os.path.join(tmpdir, f"sp{rank}"), "w") as fh:
        fh.write("1" if ok else "0")
    dist.destroy_process_group()


def test_spatial_decomposition_gloo_world2(tmp_path):
    """One periodic system, two processes, one all-reduce of forces + energy (the N > 1 path of parallel.SpatialEvaluator)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_spatial_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"sp{r}").read() == "1"


def test_spatial_evaluator_argument_checks_and_single_rank():
    """Host logic of parallel.SpatialEvaluator: a triclinic box and a halo longer than the box are refused; world size 1 hands the whole
    system through with unit weights; every atom is owned by exactly one slab and the slabs' halos hold periodic images only within
    (num_layers + 1) cutoffs of the slab."""
    from torchmdnet_amd.parallel import SpatialEvaluator

    z, pos, box = _periodic_system(200, [20.0, 8.0, 9.0], seed=1)
    ev = SpatialEvaluator(lambda *a: None, cutoff_upper=2.0, num_layers=1)
    with pytest.raises(ValueError):
        ev.local_system(pos, box + torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0], [0.0, 0.0, 0.0]]), 0, 2)
    with pytest.raises(ValueError):
        SpatialEvaluator(lambda *a: None, cutoff_upper=8.0, num_layers=2).local_system(pos, box, 0, 2)  # halo 24 A > the 20 A box
    g, p, b, n = ev.local_system(pos, box, 0, 1)
    assert n == 200 and torch.equal(p, pos) and torch.equal(b, box)
    owned = torch.zeros(200, dtype=torch.long)
    for r in range(4):
        gidx, pos_l, box_l, n_own = ev.local_system(pos, box, r, 4)
        owned[gidx[:n_own]] += 1
        x0, w, h = r * 5.0, 5.0, ev.halo
        xa = pos_l[:, 0]
        assert bool(((xa[:n_own] >= x0) & (xa[:n_own] < x0 + w)).all())
        assert bool(((xa[n_own:] >= x0 - h - 1e-5) & (xa[n_own:] < x0 + w + h + 1e-5)).all())
        assert float(box_l[0, 0]) > w + 2 * h + ev.cutoff  # more than one cutoff of vacuum between the images of the local box
        # a ghost is a periodic image of its atom
        d = torch.remainder(pos_l[n_own:, 0] - pos[gidx[n_own:], 0] + 10.0, 20.0) - 10.0
        assert float(d.abs().max()) < 1e-4
    assert bool((owned == 1).all())


def _ddp_worker(rank, world, port, tmpdir):
    """Data-parallel training: the model's parameter gradients come out of a custom autograd function (one engine pass per
    backward); wrapped in DistributedDataParallel they must be averaged over the ranks like any other gradient.  The engine calls are
    replaced by stand-ins whose gradients depend on the rank (no GPU here): what is tested is that the graph the forward builds -
    parameters as inputs of _EnergyForceParamGrad - is one DDP's hooks see."""
    import contextlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "torchmd-net_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torchmdnet_amd.models.model as M
    from torchmdnet_amd import workloads as W

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different initial weights per rank: DDP broadcasts rank 0's
    model = M.create_model(dict(W.TINY_ARGS, derivative=True))
    model.parameter_gradients = True
    model.force_gradient_order = 0
    n, n_mol = 7, 2
    z, batch = torch.ones(n, dtype=torch.long), torch.tensor([0, 0, 0, 1, 1, 1, 1])
    E0, F0 = torch.tensor([1.5, -2.0]), torch.arange(n * 3, dtype=torch.float32).reshape(n, 3) / 10
    params = [p for p in model.parameters() if p.requires_grad]
    w = float(rank + 1)
    M._require_cuda = lambda *a, **k: None
    M._direct_radial_functions = lambda m: contextlib.nullcontext()
    model.energy_and_forces = lambda *a, **k: (E0.clone(), F0.clone())
    model.parameter_gradients_of = lambda z_, p_, b_, box, q, nm, ge: (E0, {p: torch.full_like(p, w * float(ge.sum())) for p in params})

    def second(z_, p_, b_, box, q, nm, v, want_hv=False, ge=None, scale=1.0):
        # seeded (one-pass training): the gradient of s - sum ge E, i.e. the force term minus the energy pass' stand-in
        g = {p: torch.full_like(p, scale * w * (float(v.sum()) - (0.0 if ge is None else float(ge.sum())))) for p in params}
        return (g, torch.zeros(n, 3)) if want_hv else g

    model.force_term_parameter_gradients = second
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    first = params[0].detach().clone()
    gathered = [torch.zeros_like(first) for _ in range(world)]
    dist.all_gather(gathered, first)
    same_start = all(torch.equal(gathered[0], t) for t in gathered)  # rank 0's weights everywhere
    ge, R = torch.tensor([0.5, 2.0]), torch.ones(n, 3)
    pos = torch.zeros(n, 3)
    y, F = ddp(z, pos, batch)
    ((y.view(-1) * ge).sum() + (F * R).sum()).backward()
    # per rank: w * (sum ge - sum R); DDP averages over the ranks
    expect = sum((r + 1.0) for r in range(world)) / world * (float(ge.sum()) - float(R.sum()))
    ok = same_start and all(p.grad is not None and torch.allclose(p.grad, torch.full_like(p, expect), rtol=1e-5, atol=1e-5) for p in params)
    with open(os.path.join(tmpdir, f"ddp{rank}"), "w") as fh:
        fh.write("1" if ok else "0")
    dist.destroy_process_group()


def test_force_matching_gradients_average_under_ddp_gloo_world2(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"ddp{r}").read() == "1"


# ---------------------------------------------------------------------------------- per-layer halo exchange (host logic)
def _toy_pairs(pos, box, rc):
    """All directed pairs (i, j, d, unit vector from j to i) of a small orthorhombic periodic system, minimum image."""
    L = torch.diagonal(box)
    d = pos[:, None, :] - pos[None, :, :]
    d = d - torch.round(d / L) * L
    r = d.norm(dim=-1)
    i, j = torch.nonzero((r < rc) & (r > 0), as_tuple=True)
    return i, j, r[i, j], d[i, j] / r[i, j][:, None]


def _toy_step(z, pos, box, w, rc, exchange=None):
    """A two-sweep model with the structure of a TensorNet step, forward and hand-written reverse: an embedding sum over geometry and
    species, a per-atom function, a message sum over the neighbours' rows; the reverse gathers the neighbours' adjoints.  The three
    places where a sweep reads a neighbour's row are the three stages of the exchange (tmdnet_set_halo_exchange)."""
    C = 4
    s = torch.cos(z[:, None].double() * torch.arange(1, C + 1).double())          # species rows [n, C]
    i, j, r, u = _toy_pairs(pos.double(), box.double(), rc)
    phi_f = lambda t: torch.cos(t * 0.7) * (rc - t) ** 2
    dphi_f = lambda t: -0.7 * torch.sin(t * 0.7) * (rc - t) ** 2 - 2 * torch.cos(t * 0.7) * (rc - t)
    psi_f = lambda t: (rc - t) ** 3
    dpsi_f = lambda t: -3 * (rc - t) ** 2
    n = z.shape[0]
    a = torch.zeros(n, C, dtype=torch.float64).index_add_(0, i, phi_f(r)[:, None] * s[j])
    p = torch.tanh(a)
    if exchange is not None:
        exchange(0, p, None)
    m = torch.zeros(n, C, dtype=torch.float64).index_add_(0, i, psi_f(r)[:, None] * p[j])
    e = (m * p).sum(1) + (a * a).sum(1)
    E = (w.double() * e).sum()
    # reverse
    ge = w.double()[:, None]
    gm = ge * p
    if exchange is not None:
        exchange(100, gm, None)
    gp = ge * m + torch.zeros(n, C, dtype=torch.float64).index_add_(0, i, psi_f(r)[:, None] * gm[j])
    ga = gp * (1 - p * p) + 2 * a * ge
    if exchange is not None:
        exchange(200, ga, None)
    # d E / d d_ij of the directed pair (i <- j), both directions of the undirected pair appear in the list
    gd = dpsi_f(r) * (gm[i] * p[j]).sum(1) + dphi_f(r) * (ga[i] * s[j]).sum(1)
    # row i moves along u, row j against it; summing a pair's two directed entries on atom i gives its complete force
    F = torch.zeros(n, 3, dtype=torch.float64).index_add_(0, i, -gd[:, None] * u).index_add_(0, j, gd[:, None] * u)
    return E.reshape(1).float(), F.float()


def _toy_whole(z, pos, box, rc):
    return _toy_step(z, pos, box, torch.ones(z.shape[0]), rc)


def test_toy_step_reverse_is_the_gradient():
    z, pos, box = _periodic_system(40, [9.0, 5.5, 6.0], seed=4)
    pos = pos.double().requires_grad_(True)
    L = torch.diagonal(box).double()
    d = pos[:, None, :] - pos[None, :, :]
    d = d - torch.round(d.detach() / L) * L
    r = (d * d).sum(-1).add(torch.eye(40, dtype=torch.float64)).sqrt()
    mask = ((r < 2.5) & ~torch.eye(40, dtype=torch.bool)).double()
    s = torch.cos(z[:, None].double() * torch.arange(1, 5).double())
    a = ((torch.cos(r * 0.7) * (2.5 - r) ** 2 * mask)[:, :, None] * s[None, :, :]).sum(1)
    p = torch.tanh(a)
    m = (((2.5 - r) ** 3 * mask)[:, :, None] * p[None, :, :]).sum(1)
    E = ((m * p).sum(1) + (a * a).sum(1)).sum()
    (g,) = torch.autograd.grad(E, pos)
    E2, F2 = _toy_whole(z, pos.detach().float(), box, 2.5)
    assert abs(float(E.detach()) - float(E2)) < 1e-5 * abs(float(E.detach()))
    assert (F2.double() + g).abs().max().item() < 1e-5 * g.abs().max().item()


class _Mailbox:
    def __init__(self, world):
        import threading
        self.barrier, self.slots = threading.Barrier(world), {}

    def transport(self, rank, world, send, send_counts, recv_counts):
        for p, rows in enumerate(torch.split(send, send_counts)):
            self.slots[(rank, p)] = rows.clone()
        self.barrier.wait()
        got = torch.cat([self.slots[(p, rank)] for p in range(world)])
        self.barrier.wait()
        return got


def test_halo_exchange_plan_and_protocol_ranks_as_threads():
    """parallel.HaloExchangeEvaluator on the toy step: 2, 3 and 4 slabs with a halo of ONE cutoff, ghost rows refreshed at the three
    stages; every rank's owned forces are complete (no reduction over the ranks) and the energies add up.  Without the exchange
    the same decomposition is wrong, so the comparison would notice a hook that does nothing."""
    import threading
    from torchmdnet_amd.parallel import HaloExchangeEvaluator

    rc = 2.5
    for lengths, worlds in (([21.0, 6.0, 5.5], (2, 3, 4)), ([6.0, 5.5, 21.0], (3,))):  # slab axis z: the local axes are permuted
      z, pos, box = _periodic_system(90, lengths, seed=11)
      Ew, Fw = _toy_whole(z, pos, box, rc)
      for world in worlds:
          mb = _Mailbox(world)
          evs = [HaloExchangeEvaluator(lambda zl, pl, bl, wl, ex, grid: _toy_step(zl, pl, bl, wl, rc, ex), rc, transport=mb.transport)
                 for _ in range(world)]
          plans = [evs[0].plan(pos, box, r, world) for r in range(world)]
          for r in range(world):  # what r sends to p is what p expects from r, atom by atom; every ghost has exactly one source
              for p in range(world):
                  assert torch.equal(plans[r].gidx[plans[r].send[p]], plans[p].gidx[plans[p].recv[r]])
              assert sum(int(t.numel()) for t in plans[r].recv) == plans[r].n_ghost and int(plans[r].recv[r].numel()) == 0
          assert sum(pl.n_own for pl in plans) == 90
          out, err = [None] * world, []

          def run(r):
              try:
                  out[r] = evs[r].step(z, pos, box, r, world)
              except BaseException as e:  # noqa: BLE001
                  err.append(e)
                  mb.barrier.abort()

          th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
          [t.start() for t in th]
          [t.join() for t in th]
          assert not err, err
          E, F = torch.zeros(1), torch.zeros(90, 3)
          for plan, e, f_l in out:
              E = E + e
              F[plan.gidx[:plan.n_own]] = f_l[:plan.n_own]
          assert abs(float(E) - float(Ew)) < 1e-5 * abs(float(Ew)), world
          assert (F - Fw).abs().max().item() < 1e-5 * Fw.abs().max().item(), world
          assert all(ev.rows_moved == 3 * 4 * pl.n_ghost for ev, pl in zip(evs, plans))
    ev = HaloExchangeEvaluator(lambda zl, pl, bl, wl, ex, grid: _toy_step(zl, pl, bl, wl, rc, None), rc)
    plan, _, f_l = ev.step(z, pos, box, 0, 2)
    assert (f_l[:plan.n_own] - Fw[plan.gidx[:plan.n_own]]).abs().max().item() > 1e-2 * Fw.abs().max().item()
    squeezed = pos.clone()
    squeezed[:, 2] = squeezed[:, 2] * 0.3  # all atoms in the lowest third of the slab axis: the upper slabs are vacuum
    with pytest.raises(ValueError, match="without atoms"):
        HaloExchangeEvaluator(None, rc).plan(squeezed, box, 0, 3)
    with pytest.raises(ValueError, match="own ghost"):
        HaloExchangeEvaluator(None, 11.0).plan(pos, box, 0, 2)  # a halo of 11 A on either side of a 10.5 A slab in 21 A


def test_halo_plan_atoms_on_and_beside_slab_faces():
    """ownership and ghost membership come from one integer per atom, so an atom on a slab face, a few ulp
    beside it, or an ulp below 0 (where the wrap rounds to the box length) is local to every rank that owns one of its
    neighbours.  Box lengths and worlds include the advisor's failing cases (x = 31.049997 in 62.1 with 6 slabs, 82.75 in 99.3)."""
    import numpy as np
    from torchmdnet_amd.parallel import HaloExchangeEvaluator

    rc = 5.0
    for La, world in ((62.1, 6), (99.3, 6), (40.0, 4), (31.0, 2), (30.0, 6), (61.7, 3)):
        lengths = torch.tensor([La, 11.0, 12.0])
        box = torch.diag(lengths)
        g = torch.Generator().manual_seed(int(La * 10) + world)
        pos = torch.rand(600, 3, generator=g) * lengths
        w = La / world
        xs = [31.049997, 82.75, -1e-7, -1e-9, 0.0, La, float(np.nextafter(np.float32(La), np.float32(0)))]
        for p in range(world + 1):
            f32 = np.float32(p * w)
            xs += [float(f32)]
            up, dn = f32, f32
            for _ in range(3):
                up, dn = np.nextafter(up, np.float32(1e9)), np.nextafter(dn, np.float32(-1e9))
                xs += [float(up), float(dn)]
            n_w = int(w // rc)
            if n_w >= 1:  # the far side of the halo cell too
                xs += [float(np.float32(p * w + w / n_w)), float(np.float32(p * w - w / n_w))]
        face = torch.rand(len(xs), 3, generator=g) * lengths
        face[:, 0] = torch.tensor(xs, dtype=torch.float32)
        pos = torch.cat([pos, face])
        n = pos.shape[0]
        ev = HaloExchangeEvaluator(None, rc)
        plans = [ev.plan(pos, box, r, world) for r in range(world)]
        owned = torch.cat([pl.gidx[:pl.n_own] for pl in plans])
        assert sorted(owned.tolist()) == list(range(n))  # every atom owned exactly once
        d = pos[:, None, :] - pos[None, :, :]
        d = d - lengths * torch.round(d / lengths)
        within = d.norm(dim=-1) < rc
        for r, pl in enumerate(plans):
            local = torch.zeros(n, dtype=torch.bool)
            local[pl.gidx] = True
            assert pl.gidx.unique().numel() == pl.gidx.numel()
            need = within[pl.gidx[:pl.n_own]].any(dim=0)
            missing = torch.nonzero(need & ~local).flatten()
            assert missing.numel() == 0, (La, world, r, pos[missing, 0].tolist())
            for p in range(world):
                assert torch.equal(pl.gidx[pl.send[p]], plans[p].gidx[plans[p].recv[r]])
            assert sum(int(t.numel()) for t in pl.recv) == pl.n_ghost


def test_halo_exchange_enters_a_collective_transport_with_zero_rows():
    """a rank without ghost traffic still enters a collective transport (its peers are in the all-to-all)."""
    from torchmdnet_amd.parallel import HaloExchangeEvaluator, HaloPlan

    calls = []

    def transport(rank, world, send, send_counts, recv_counts):
        calls.append((tuple(send.shape), tuple(send_counts), tuple(recv_counts)))
        return send.new_zeros((0, send.shape[1]))

    none = torch.zeros(0, dtype=torch.long)
    plan = HaloPlan(torch.arange(4), torch.zeros(4, 3), torch.eye(3), 4, [none, none], [none, none])
    rows = torch.ones(4, 6)
    ev = HaloExchangeEvaluator(None, 1.0, transport=transport)
    ev.exchange_fn(plan, 0, 2)(0, rows, None)
    assert calls == []  # a point-to-point transport: nothing to do
    transport.collective = True
    ev.exchange_fn(plan, 0, 2)(0, rows, None)
    assert calls == [((0, 6), (0, 0), (0, 0))] and bool((rows == 1).all())


def _halo_worker(rank, world, port, tmpdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "torchmd-net_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from torchmdnet_amd.parallel import HaloExchangeEvaluator

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rc = 2.5
    z, pos, box = _periodic_system(90, [21.0, 6.0, 5.5], seed=11)
    ev = HaloExchangeEvaluator(lambda zl, pl, bl, wl, ex, grid: _toy_step(zl, pl, bl, wl, rc, ex), rc)
    E, F = ev.evaluate(z, pos, box)  # all-to-all per exchange, then the [3 N + 1] all-reduce
    Ew, Fw = _toy_whole(z, pos, box, rc)
    ok = abs(float(E) - float(Ew)) < 1e-5 * abs(float(Ew)) and (F - Fw).abs().max().item() < 1e-5 * Fw.abs().max().item()
    ok = ok and ev.rows_moved > 0
    with open(os.path.join(tmpdir, f"hx{rank}"), "w") as fh:
        fh.write("1" if ok else "0")
    dist.destroy_process_group()


def test_halo_exchange_gloo_world2(tmp_path):
    """The default transport: one all_to_all_single per exchange between two processes (gloo here, RCCL on GPUs)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_halo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"hx{r}").read() == "1"
