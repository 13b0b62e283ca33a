"""Multi-GPU evaluation: molecules are independent graphs (the pair search never crosses a molecule,
reference warp_kernels/neighbors_brute.py:107-110), so a batch shards by molecule with NO data-path
collective for the forces; the only exchange is the per-molecule energy vector (RCCL all-reduce of a
zero-padded [n_mol] fp32 vector over xGMI: 1 KiB for 256 molecules, latency-bound).

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).
The compute callable is injected so that the partition / reduction logic is testable without a GPU.
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def molecule_ranges(batch: torch.Tensor, n_mol: int, world: int):
    """Contiguous molecule ranges balanced by atom count (edges ~ atoms).  ``batch`` must be sorted.
    Returns a list of (mol_lo, mol_hi, atom_lo, atom_hi), one per rank (possibly empty ranges)."""
    counts = torch.bincount(batch.cpu(), minlength=n_mol)
    csum = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        # first molecule boundary whose atom offset is >= target
        m = int(torch.searchsorted(csum, torch.tensor(target, dtype=csum.dtype)))
        m = min(max(m, bounds[-1]), n_mol)
        bounds.append(m)
    bounds.append(n_mol)
    return [(bounds[r], bounds[r + 1], int(csum[bounds[r]]), int(csum[bounds[r + 1]])) for r in range(world)]


class ShardedEvaluator:
    """``evaluate(z, pos, batch, q=None, box=None, n_mol=None)`` on every rank with the FULL batch:
    each rank computes its molecule range, energies are summed across ranks, forces stay local
    (``forces_local`` covers atoms [atom_lo, atom_hi)).  ``gather_forces=True`` additionally
    all-gathers the force blocks (variable sizes are padded to the largest block)."""

    def __init__(self, compute: Callable, group: Optional[dist.ProcessGroup] = None, gather_forces: bool = False):
        self.compute = compute
        self.group = group
        self.gather_forces = gather_forces

    def plan(self, batch, n_mol=None):
        """Partition of a (sorted) batch vector over the ranks: list of (mol_lo, mol_hi, atom_lo, atom_hi).  It depends on
        `batch` only, so an MD loop or a benchmark computes it once and passes it to evaluate(ranges=...): the check and the
        partition read `batch` on the host (one synchronisation)."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if n_mol is None:
            n_mol = int(batch.max().item()) + 1
        if not bool((batch[1:] >= batch[:-1]).all()):
            raise ValueError("sharded evaluation needs atoms of a molecule to be contiguous (sorted batch)")
        return molecule_ranges(batch, n_mol, world)

    def evaluate(self, z, pos, batch, q=None, box=None, n_mol=None, ranges=None) -> Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        if n_mol is None:
            n_mol = int(batch.max().item()) + 1
        if ranges is None:
            ranges = self.plan(batch, n_mol)
        m_lo, m_hi, a_lo, a_hi = ranges[rank]
        energy = torch.zeros(n_mol, dtype=torch.float32, device=pos.device)
        if m_hi > m_lo:
            zl, pl = z[a_lo:a_hi], pos[a_lo:a_hi]
            bl = batch[a_lo:a_hi] - m_lo
            ql = None if q is None else q[m_lo:m_hi]
            boxl = box if (box is None or box.dim() == 2) else box[m_lo:m_hi]
            e_loc, f_loc = self.compute(zl, pl, bl, boxl, ql, m_hi - m_lo)
            energy[m_lo:m_hi] = e_loc.reshape(-1)
        else:
            f_loc = torch.zeros((0, 3), dtype=torch.float32, device=pos.device)
        if world > 1:
            dist.all_reduce(energy, op=dist.ReduceOp.SUM, group=self.group)
        if self.gather_forces and world > 1:
            mx = max(r[3] - r[2] for r in ranges)
            pad = torch.zeros((mx, 3), dtype=torch.float32, device=pos.device)
            pad[: f_loc.shape[0]] = f_loc
            blocks = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(blocks, pad, group=self.group)
            f_all = torch.cat([blocks[r][: ranges[r][3] - ranges[r][2]] for r in range(world)])
            return energy, f_all, (0, pos.shape[0])
        return energy, f_loc, (a_lo, a_hi)

    def prepare(self, z, pos, batch, q=None, box=None, n_mol=None, ranges=None, graph=False, guard=None, warmup=3):
        """Stepping form of ``evaluate`` for MD loops and benchmarks: the rank's shard is cut out ONCE, the outputs are static
        buffers, and with ``graph=True`` the local evaluation (+ the zero-padding of the energy vector) is captured into one
        HIP graph, so a step is `graph.replay()` + one RCCL all-reduce - no slicing, no allocation, no host synchronisation
        (the capture pattern of the reference's TorchMD adapter, calculators.py:117-128).  ``compute`` must then be
        capture-safe (a model created with static_shapes=True).  Returns a :class:`ShardSession`."""
        if n_mol is None:
            n_mol = int(batch.max().item()) + 1
        if ranges is None:
            ranges = self.plan(batch, n_mol)
        return ShardSession(self, z, pos, batch, q, box, n_mol, ranges, graph, guard, warmup)

    def prepare_local(self, z_l, pos_l, batch_l, n_mol, mol_lo, atom_lo=0, q_l=None, box_l=None, graph=False, guard=None, warmup=3):
        """``prepare`` for a rank that holds ONLY its own molecules (weak scaling, data-parallel inference over a stream of
        molecules): ``batch_l`` counts from 0, the rank's energies land in slots [mol_lo, mol_lo + n_local) of the global
        zero-padded vector of ``n_mol`` entries."""
        n_loc = int(batch_l.max().item()) + 1 if batch_l.numel() else 0
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        ranges = [None] * world
        ranges[rank] = (mol_lo, mol_lo + n_loc, atom_lo, atom_lo + int(z_l.shape[0]))
        return ShardSession(self, z_l, pos_l, batch_l + mol_lo, q_l, box_l, n_mol, ranges, graph, guard, warmup, local=True)


class ShardSession:
    """One rank's prepared shard (see ShardedEvaluator.prepare).  ``step(new_pos=None)`` returns
    ``(energy [n_mol] summed over the ranks, forces of the local atoms, (atom_lo, atom_hi))``; the tensors are static buffers
    that the next step overwrites."""

    def __init__(self, ev, z, pos, batch, q, box, n_mol, ranges, graph, guard, warmup, local=False):
        self.ev, self.n_mol, self.ranges = ev, n_mol, ranges
        self.world = dist.get_world_size(ev.group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(ev.group) if dist.is_initialized() else 0
        m_lo, m_hi, a_lo, a_hi = ranges[self.rank]
        self.m_lo, self.m_hi, self.a_lo, self.a_hi = m_lo, m_hi, a_lo, a_hi
        dev = pos.device
        self.energy = torch.zeros(n_mol, dtype=torch.float32, device=dev)
        sl_a = slice(None) if local else slice(a_lo, a_hi)  # local: the inputs already are this rank's rows
        sl_m = slice(None) if local else slice(m_lo, m_hi)
        self.local = local
        # in the dtypes the engine takes: a captured graph records the pointers of exactly these tensors (a conversion inside the
        # compute callable would hand the graph a temporary)
        self.z_l = z[sl_a].to(torch.long).contiguous()
        self.pos_l = pos[sl_a].detach().to(torch.float32).clone().contiguous()
        self.batch_l = (batch[sl_a] - m_lo).to(torch.long).contiguous()
        self.q_l = None if q is None else q[sl_m].detach().to(device=dev, dtype=torch.float32).contiguous()
        self.box_l = None if box is None else (box if box.dim() == 2 else box[sl_m]).detach().to(device=dev, dtype=torch.float32).contiguous()
        self.f_loc = torch.zeros((a_hi - a_lo, 3), dtype=torch.float32, device=dev)
        self.graph, self.guard, self._token = None, guard, None
        if graph and m_hi > m_lo:
            if not pos.is_cuda:
                raise RuntimeError("graph=True needs device tensors")
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(warmup, 1)):
                    self._local()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._local()
            self._token = guard() if guard else None

    def _local(self):
        self.energy.zero_()
        if self.m_hi > self.m_lo:
            e_loc, f_loc = self.ev.compute(self.z_l, self.pos_l, self.batch_l, self.box_l, self.q_l, self.m_hi - self.m_lo)
            self.energy[self.m_lo:self.m_hi] = e_loc.reshape(-1)
            self.f_loc = f_loc

    def step(self, new_pos=None):
        if new_pos is not None:  # full-system positions: the rank keeps its own rows
            src = (new_pos if self.local else new_pos[self.a_lo:self.a_hi]).detach()
            if self.pos_l.is_cuda:
                torch.mul(src, 1.0, out=self.pos_l)  # an elementwise kernel, not the runtime's blit path (see TorchMD_Net.capture)
            else:
                self.pos_l.copy_(src)
        if self.graph is not None:
            if self.guard and self.guard() != self._token:
                raise RuntimeError("stale HIP graph: the model's parameters or workspaces changed after prepare(); prepare again")
            self.graph.replay()
        else:
            self._local()
        if dist.is_initialized():  # also at world size 1: the collective path is the same code on every node size
            dist.all_reduce(self.energy, op=dist.ReduceOp.SUM, group=self.ev.group)
        return self.energy, self.f_loc, (self.a_lo, self.a_hi)


class SpatialEvaluator:
    """ONE large periodic system over the ranks (last paragraph; the reference has no counterpart: a
    system must fit one GPU there).  First executable step of the spatial decomposition: slabs along one axis of an
    orthorhombic box, DEEP halo, no exchange inside the step.

    The energy of an atom depends on the positions within (num_layers + 1) * cutoff_upper of it: one cutoff for the
    embedding's neighbour sum and one per interaction layer (reference tensornet.py:543-619, 757-806).  A rank therefore takes
    the atoms of its slab (weight 1 in the energy sum) plus every periodic image within that distance of the slab as ghost copies
    (weight 0), evaluates this local system as an ordinary periodic system - the local box is the slab + both halos + more than
    one cutoff of vacuum along the slab axis, so the images of the local box do not see each other; the other two axes keep the
    box's own periodicity - and gets E_r = sum of its owned atoms' energies and -dE_r/d(pos) on owned atoms AND ghosts
    (``tmdnet_set_atom_weights``).  The forces of the whole system are the sum over ranks and copies, so the only collective is
    ONE all-reduce of [3N + 1] floats per step (forces + energy); positions are replicated like every other input.  The price
    is the redundant evaluation of the halo: it pays when the slab is wide against 2 (L + 1) rc (30 Angstrom for the C2
    model); the per-layer halo exchange that removes the redundancy is the next step.

    ``compute(z_l, pos_l, box_l, w_l) -> (E [1], F_l [n_l, 3])`` is injected (the engine on a GPU, the oracle in the CPU tests);
    ``energy_offset`` is the model's per-molecule ``mean``, which every rank's E contains once."""

    def __init__(self, compute: Callable, cutoff_upper: float, num_layers: int, group: Optional[dist.ProcessGroup] = None,
                 axis: Optional[int] = None, energy_offset: float = 0.0):
        self.compute, self.group, self.axis = compute, group, axis
        self.cutoff = float(cutoff_upper)
        self.halo = (int(num_layers) + 1) * float(cutoff_upper)
        self.energy_offset = float(energy_offset)

    def local_system(self, pos: torch.Tensor, box: torch.Tensor, rank: int, world: int):
        """-> (gidx [n_l] global index of every local atom, owned atoms first; pos_l [n_l, 3]; box_l [3, 3]; n_owned)."""
        if box.dim() != 2 or bool((box - torch.diag(torch.diagonal(box))).abs().max() > 0):
            raise ValueError("SpatialEvaluator: one orthorhombic box [3, 3] (diagonal) for the whole system")
        lengths = torch.diagonal(box)
        a = int(torch.argmax(lengths)) if self.axis is None else int(self.axis)
        La = float(lengths[a])
        if world == 1:
            return torch.arange(pos.shape[0], device=pos.device), pos, box, pos.shape[0]
        h, w = self.halo, La / world
        if h > La:
            raise ValueError(f"halo {h:g} exceeds the box length {La:g} along the slab axis: more than one image per atom")
        x0 = rank * w
        x = torch.remainder(pos[:, a], La)
        s = torch.clamp(torch.floor(x / w).long(), max=world - 1)  # slab of every atom (x == La rounds into the last one)
        owned = s == rank
        dlo = torch.remainder(x0 - x, La)          # distance below the slab's lower face (periodic)
        dhi = torch.remainder(x - (x0 + w), La)    # distance above its upper face
        # owned atoms have dlo in (La - w, La] and dhi in [La - w, La): they enter a halo only as periodic images of themselves
        lo = (dlo > 0) & (dlo <= h)
        hi = dhi < h
        i_own = torch.nonzero(owned).flatten()
        i_lo = torch.nonzero(lo).flatten()
        i_hi = torch.nonzero(hi).flatten()
        gidx = torch.cat([i_own, i_lo, i_hi])
        pos_l = pos[gidx].clone()
        # local coordinate along the slab axis = the atom's own (wrapped) coordinate, or its periodic image one box length away:
        # no translation, so the distances inside the local system are rounded like the whole system's wherever they can be
        x1 = x0 + w
        xa = torch.cat([x[i_own], torch.where(x[i_lo] < x0, x[i_lo], x[i_lo] - La), torch.where(x[i_hi] >= x1, x[i_hi], x[i_hi] + La)])
        pos_l[:, a] = xa
        box_l = box.clone()
        box_l[a, a] = w + 2 * h + 1.05 * self.cutoff + 1e-3  # [x0 - h, x1 + h) plus more than one cutoff of vacuum
        return gidx, pos_l, box_l, int(i_own.numel())

    def contribution(self, z, pos, box, rank: int, world: int):
        """This rank's term of the sums: (E_r - offset [1], forces scattered to the global atom order [N, 3])."""
        gidx, pos_l, box_l, n_own = self.local_system(pos, box, rank, world)
        forces = torch.zeros((pos.shape[0], 3), dtype=torch.float32, device=pos.device)
        if n_own == 0:
            return torch.zeros(1, dtype=torch.float32, device=pos.device), forces
        w_l = torch.zeros(gidx.numel(), dtype=torch.float32, device=pos.device)
        w_l[:n_own] = 1.0
        e, f_l = self.compute(z[gidx], pos_l, box_l, w_l)
        forces.index_add_(0, gidx, f_l.to(torch.float32))
        return e.reshape(1).to(torch.float32) - self.energy_offset, forces

    def evaluate(self, z, pos, box) -> Tuple[torch.Tensor, torch.Tensor]:
        """Every rank calls with the full (replicated) system; returns (E [1], F [N, 3]) of the whole system on every rank."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        e, forces = self.contribution(z, pos, box, rank, world)
        if world > 1:
            buf = torch.cat([forces.reshape(-1), e])
            if buf.is_cuda and dist.get_backend(self.group) == "gloo":  # tests: two processes on one GPU (RCCL refuses that)
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                buf = host.to(buf.device)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            forces, e = buf[:-1].reshape(-1, 3), buf[-1:]
        return e + self.energy_offset, forces


class HaloPlan:
    """What one rank needs for a step of ``HaloExchangeEvaluator``: its local system and, per peer, the local rows it sends
    (owned atoms that are ghosts over there) and the local rows it receives into (its ghosts owned over there), both in the order
    of the receiver's ghost list.  ``order``: the axes of the local system (slab axis first); ``cell_grid``: the grid of the
    engine's neighbour search that puts the owned atoms into one contiguous range of its cell order, or None."""

    def __init__(self, gidx, pos_l, box_l, n_own, send, recv, order=(0, 1, 2), cell_grid=None):
        self.gidx, self.pos_l, self.box_l, self.n_own, self.send, self.recv = gidx, pos_l, box_l, n_own, send, recv
        self.order, self.cell_grid = tuple(order), cell_grid

    @property
    def n_ghost(self) -> int:
        return int(self.gidx.numel()) - self.n_own


def distributed_transport(group: Optional[dist.ProcessGroup] = None):
    """Rows between the ranks by ONE all-to-all per exchange (RCCL over xGMI; under gloo - the tests, two processes on one GPU -
    the device rows are staged through the host).  ``transport(rank, world, send, send_counts, recv_counts) -> recv``: ``send``
    [sum(send_counts), width] holds the rows for peer 0, then peer 1, ...; ``recv`` [sum(recv_counts), width] likewise."""

    def transport(rank: int, world: int, send: torch.Tensor, send_counts, recv_counts):
        host = send.is_cuda and dist.get_backend(group) == "gloo"
        src = send.cpu() if host else send.contiguous()
        dst = torch.empty((sum(recv_counts), send.shape[1]), dtype=src.dtype, device=src.device)
        dist.all_to_all_single(dst, src, list(recv_counts), list(send_counts), group=group)
        return dst.to(send.device) if host else dst

    transport.collective = True
    return transport


class HaloExchangeEvaluator:
    """ONE large periodic system over the ranks with a halo of ONE cell (>= one cutoff) and an exchange of ghost rows inside the step
    (the step after ``SpatialEvaluator``'s deep halo; the reference has no counterpart).

    A TensorNet step is per-atom kernels (local to a row) cut by neighbour sweeps that gather the rows of an atom's neighbours
    (reference tensornet.py:757-806: the message sum of every layer; tensornet.py:543-619: the embedding sum, which gathers
    geometry and species only).  A rank holds its slab's atoms (weight 1 in the energy sum) and the atoms within one cell of the
    slab as ghosts (weight 0), in the true periodic box: every neighbour of an owned atom is local.  What the ghosts lack is their own
    neighbourhood, so their rows are wrong wherever a sweep has been - and are replaced, before each sweep that gathers them, by
    the rows their owners computed: P_l before the message sum of layer l, the adjoint of that sum's output on the way back,
    and the adjoint of the embedding sum before the pair gradients of the embedding; 2 L + 1 exchanges of [n_ghost, 9 F or 10 F]
    floats (``tmdnet_set_halo_exchange``, include/tmdnet_amd.h).  The forces of the owned atoms come out complete - an owned
    atom's pairs are all local and both halves of a pair's distance gradient are computed from exchanged rows - so, unlike the
    deep halo, nothing is reduced over the ranks: every rank returns the forces of its own atoms and the energy is one scalar
    all-reduce.  ``evaluate`` still all-reduces a [3 N + 1] vector so that every rank ends with the whole answer, like
    ``SpatialEvaluator.evaluate``; an MD driver that keeps the atoms distributed uses ``step`` and skips it.

    ``compute(z_l, pos_l, box_l, w_l, exchange, cell_grid) -> (E [1], F_l [n_l, 3])`` is injected (the engine:
    ``model.energy_and_forces(..., atom_weights=w_l, halo_exchange=exchange, cell_grid=cell_grid)``);
    ``transport(rank, world, send, send_counts, recv_counts) -> recv`` moves the rows (``distributed_transport``; the tests also run the ranks as threads of one process with a mailbox)."""

    def __init__(self, compute: Callable, cutoff_upper: float, group: Optional[dist.ProcessGroup] = None,
                 axis: Optional[int] = None, energy_offset: float = 0.0, transport: Optional[Callable] = None):
        self.compute, self.group = compute, group
        self.energy_offset = float(energy_offset)
        self.cutoff = float(cutoff_upper)
        self.axis = axis
        self.transport = transport if transport is not None else distributed_transport(group)
        self.rows_moved = 0  # floats received by this rank in the last step (for the probes)

    @classmethod
    def for_model(cls, model, group: Optional[dist.ProcessGroup] = None, axis: Optional[int] = None, transport: Optional[Callable] = None,
                  q: Optional[torch.Tensor] = None):
        """The evaluator of a ``torchmdnet_amd`` TensorNet model on its GPU: cutoff and energy offset from the model, the engine call
        (``energy_and_forces`` with the atom weights, the exchange and the aligned cell grid) as ``compute``."""

        def compute(z_l, pos_l, box_l, w_l, exchange, cell_grid):
            return model.energy_and_forces(z_l, pos_l, torch.zeros_like(z_l), box_l, q, 1, True, atom_weights=w_l, halo_exchange=exchange,
                                           cell_grid=cell_grid)

        return cls(compute, float(model.representation_model.cutoff_upper), group, axis, float(model.mean), transport)

    def plan(self, pos: torch.Tensor, box: torch.Tensor, rank: int, world: int) -> HaloPlan:
        """Computed by every rank from the replicated positions: no negotiation, O(N) work.

        The local system keeps the TRUE box and the atoms' own coordinates: the ghosts are the atoms (not images) within one halo
        width of the slab along the periodic slab axis, each once, and the engine's minimum-image arithmetic is the whole system's -
        no shifted copies whose coordinates would round differently (a shift by a 99 A box length moves fp32 coordinates by up to
        4e-6 A, 1e-5 of the forces next to the cut).  Pairs of two ghosts may appear or not (the engine leaves them out): nothing
        of a ghost's own neighbourhood is used.

        The slabs are aligned with the engine's cell grid: a slab of width w holds n_w = floor(w / cutoff) cells of width
        c = w / n_w >= cutoff along the slab axis, the halo is ONE such cell on either side, and the local system's axes are
        permuted so that the slab axis comes first - the major key of the engine's cell order.  In cell order the owned atoms are
        then one contiguous range, and the engine runs its per-atom kernels and forward sweeps on that range only
        (tmdnet_get_info "halo_active_rows"); any other arrangement is still exact, with the ghosts' rows computed in vain."""
        rc = self.cutoff
        n = pos.shape[0]
        if world == 1:
            return HaloPlan(torch.arange(n, device=pos.device), pos, box, n, [pos.new_zeros(0, dtype=torch.long)],
                            [pos.new_zeros(0, dtype=torch.long)])
        if box.dim() != 2 or bool((box - torch.diag(torch.diagonal(box))).abs().max() > 0):
            raise ValueError("HaloExchangeEvaluator: one orthorhombic box [3, 3] (diagonal) for the whole system")
        lengths = torch.diagonal(box)
        a = int(torch.argmax(lengths)) if self.axis is None else int(self.axis)
        La = float(lengths[a])
        w = La / world
        n_w = int(w // rc)
        h = w / n_w if n_w >= 1 else rc            # halo = one cell >= one cutoff; slabs thinner than a cutoff: no alignment
        if 2 * h > La - w:
            raise ValueError(f"two halos of {h:g} do not fit beside a slab of {w:g} in the box length {La:g}: an atom would be its "
                             "own ghost, or a ghost on both sides")
        # Ownership and halo membership come from ONE integer per atom - its cell along the slab axis - so that an atom on (or an
        # ulp beside) a slab face is never owned by one slab and missed as a ghost by its neighbour; the distance tests below are
        # closed, signed (an atom binned across a face by rounding has a slightly negative distance) and tolerant, and only ADD ghosts.
        x = torch.remainder(pos[:, a], La)
        x = torch.where(x >= La, torch.zeros_like(x), x)  # remainder of a tiny negative coordinate rounds to La: the same point as 0
        if n_w >= 1:
            n_cell = n_w * world
            cell = torch.clamp(torch.floor(x * (n_cell / La)).long(), 0, n_cell - 1)
            slab = torch.div(cell, n_w, rounding_mode="floor")
        else:
            n_cell, cell = 0, None
            slab = torch.clamp(torch.floor(x / w).long(), 0, world - 1)
        if int(torch.bincount(slab, minlength=world).min()) == 0:
            # every rank sees it (replicated positions) and raises: a rank without a step would leave its peers waiting in the exchanges
            raise ValueError("HaloExchangeEvaluator: a slab without atoms (vacuum along the slab axis); choose another axis or fewer ranks")
        tol = 1e-5 * h + 4e-7 * La

        def halo_of(p: int, xs: torch.Tensor, slabs: torch.Tensor, cells: Optional[torch.Tensor]):
            """Atoms (of `xs`) of another slab within h of a face of slab p: in the cell layer beside the face, or by distance."""
            dlo = p * w - xs                                 # signed distance below the slab's lower face, wrapped to [-La/2, La/2]
            dlo = dlo - La * torch.round(dlo / La)
            dhi = xs - (p * w + w)                           # signed distance above its upper face
            dhi = dhi - La * torch.round(dhi / La)
            near = ((dlo >= -tol) & (dlo <= h + tol)) | ((dhi >= -tol) & (dhi <= h + tol))
            if cells is not None:
                near = near | (cells == (p * n_w - 1) % n_cell) | (cells == ((p + 1) * n_w) % n_cell)
            return (slabs != p) & near

        own = torch.nonzero(slab == rank).flatten()
        ghosts = torch.nonzero(halo_of(rank, x, slab, cell)).flatten()  # ascending atom index, every atom once
        gidx = torch.cat([own, ghosts])
        n_own = int(own.numel())
        # a halo no wider than a slab reaches the two adjacent slabs - and, through the closed tests, an atom exactly on the far face of
        # one: every other slab is asked (world masks over the owned atoms; empty lists for all but two or three of them)
        peers = set(range(world)) - {rank}
        none = own[:0]
        g_slab = slab[ghosts]
        recv = [n_own + torch.nonzero(g_slab == p).flatten() if p in peers else none for p in range(world)]
        # what the others hold of mine, in THEIR ghost order (ascending atom index = ascending row of mine)
        x_own, s_own = x[own], slab[own]
        c_own = cell[own] if cell is not None else None
        rows = torch.arange(n_own, device=pos.device)
        send = [rows[halo_of(p, x_own, s_own, c_own)] if p in peers else none for p in range(world)]
        order = [a] + [k for k in range(3) if k != a]
        grid = None
        if n_w >= 1:
            cells = (n_w * world,) + tuple(max(1, int(float(lengths[k]) // rc)) for k in order[1:])
            if cells[0] * cells[1] * cells[2] <= 4 * int(gidx.numel()):
                grid = cells
        pos_l = pos[gidx][:, order].contiguous()
        box_l = torch.diag(lengths[order]).to(box.dtype).contiguous()
        return HaloPlan(gidx, pos_l, box_l, n_own, send, recv, order, grid)

    def exchange_fn(self, plan: HaloPlan, rank: int, world: int) -> Callable:
        send_all = torch.cat(plan.send)
        recv_all = torch.cat(plan.recv)
        send_counts = [int(t.numel()) for t in plan.send]
        recv_counts = [int(t.numel()) for t in plan.recv]
        self.rows_moved = 0

        def exchange(stage: int, rows: torch.Tensor, inv: Optional[torch.Tensor]):
            if recv_all.numel() == 0 and send_all.numel() == 0 and not getattr(self.transport, "collective", False):
                return  # a collective transport is entered by every rank every time (zero rows here): the peers would hang otherwise
            src = send_all if inv is None else inv[send_all]
            dst = recv_all if inv is None else inv[recv_all]
            rows[dst] = self.transport(rank, world, rows[src], send_counts, recv_counts)
            self.rows_moved += int(dst.numel()) * int(rows.shape[1])

        return exchange

    def step(self, z, pos, box, rank: int, world: int):
        """-> (plan, E_r - offset [1], forces of the local atoms [n_l, 3]; rows [:plan.n_own] are complete)."""
        plan = self.plan(pos, box, rank, world)
        w_l = torch.zeros(plan.gidx.numel(), dtype=torch.float32, device=pos.device)
        w_l[:plan.n_own] = 1.0
        e, f_l = self.compute(z[plan.gidx], plan.pos_l, plan.box_l, w_l, self.exchange_fn(plan, rank, world) if world > 1 else None,
                              plan.cell_grid)
        back = [plan.order.index(k) for k in range(3)]  # the local system's axes back to the caller's
        return plan, e.reshape(1).to(torch.float32) - self.energy_offset, f_l.to(torch.float32)[:, back]

    def evaluate(self, z, pos, box) -> Tuple[torch.Tensor, torch.Tensor]:
        """Every rank calls with the full (replicated) system; returns (E [1], F [N, 3]) of the whole system on every rank."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        plan, e, f_l = self.step(z, pos, box, rank, world)
        forces = torch.zeros((pos.shape[0], 3), dtype=torch.float32, device=pos.device)
        forces[plan.gidx[:plan.n_own]] = f_l[:plan.n_own]
        if world > 1:
            buf = torch.cat([forces.reshape(-1), e])
            if buf.is_cuda and dist.get_backend(self.group) == "gloo":
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                buf = host.to(buf.device)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            forces, e = buf[:-1].reshape(-1, 3), buf[-1:]
        return e + self.energy_offset, forces
