"""torch custom-op registration of the HIP path.

The reference exposes its native kernels as ``torch.library.custom_op``s with ``register_fake`` / ``register_autograd``
(torchmdnet/extensions/warp_ops/neighbors.py:34-148, warp_ops/tensornet_mp.py:538-548, entry binding
torchmdnet/models/tensornet.py:54-66); that is what lets ``torch.compile`` (calculators.py:297), ``torch.export``
(tests/test_export.py) and the CUDA-graph tests trace through them.  Here the seam is coarser - the whole energy+force
evaluation is ONE library call - so two ops are registered:

  tmdnet::energy_forces(z, pos, batch, box?, q?, engine, n_mol, want_forces) -> (energy [n_mol], forces [N,3] or [0,3])
      autograd: d(sum_m g_m E_m)/d pos_i = -g_{batch_i} F_i  (molecules are independent); no second derivatives
  tmdnet::neighbor_pairs(pos, batch, box?, cutoff_lower, cutoff_upper, max_num_pairs, loop, include_transpose, strategy)
      -> (neighbors [2,M] int64, deltas [M,3], distances [M], num_pairs [1] int32)      (reference op: same outputs)
      float32 or float64 positions, like the reference's (tmdnet_neighbor_pairs / tmdnet_neighbor_pairs_f64)
      autograd: neighbor_grad_positions of the reference (extensions/neighbor_utils.py:11-46) as a HIP kernel
      (tmdnet::neighbor_grad), itself differentiable (second derivatives: gradgradcheck of the reference's tests)

``engine`` is an integer key into a registry of TorchMD_Net modules (custom-op arguments must be tensors or scalars; the
parameters live in the library handle the module owns).  Both ops are device_types="cuda": there is no CPU kernel.
"""
import ctypes as C
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from torchmdnet_amd import _C

_ENGINES = weakref.WeakValueDictionary()
_next_id = [0]


def register_engine(module) -> int:
    _next_id[0] += 1
    _ENGINES[_next_id[0]] = module
    return _next_id[0]


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------------------------------------ energy + forces
@torch.library.custom_op("tmdnet::energy_forces", mutates_args=(), device_types="cuda")
def energy_forces(z: Tensor, pos: Tensor, batch: Tensor, box: Optional[Tensor], q: Optional[Tensor], engine: int, n_mol: int,
                  want_forces: bool) -> Tuple[Tensor, Tensor]:
    module = _ENGINES.get(engine)
    if module is None:
        raise RuntimeError(f"tmdnet::energy_forces: engine {engine} is gone (its TorchMD_Net module was deleted)")
    e, f = module.energy_and_forces(z, pos, batch, box, q, n_mol, want_forces=want_forces)
    return e, (f if f is not None else pos.new_empty((0, 3)))


@energy_forces.register_fake
def _(z, pos, batch, box, q, engine, n_mol, want_forces):
    n = pos.shape[0] if want_forces else 0
    return pos.new_empty((n_mol,), dtype=torch.float32), pos.new_empty((n, 3), dtype=torch.float32)


def _ef_setup(ctx, inputs, output):
    z, pos, batch, box, q, engine, n_mol, want_forces = inputs
    energy, forces = output
    ctx.have_forces = bool(want_forces)
    ctx.save_for_backward(forces, batch)


def _ef_backward(ctx, g_energy, g_forces):
    forces, batch = ctx.saved_tensors
    if not ctx.have_forces:
        raise RuntimeError("tmdnet::energy_forces was evaluated with want_forces=False: no position gradient is available")
    g_pos = None
    if g_energy is not None:
        g_pos = -forces * g_energy.reshape(-1)[batch.to(torch.long)].unsqueeze(-1)
    # g_forces (second derivatives, force-matching training) is not implemented on the HIP path
    return None, g_pos, None, None, None, None, None, None


energy_forces.register_autograd(_ef_backward, setup_context=_ef_setup)


# ------------------------------------------------------------------------------------------------ neighbour pairs
_ws_cache = {}


@torch.library.custom_op("tmdnet::neighbor_pairs", mutates_args=(), device_types="cuda")
def neighbor_pairs(pos: Tensor, batch: Tensor, box: Optional[Tensor], cutoff_lower: float, cutoff_upper: float, max_num_pairs: int,
                   loop: bool, include_transpose: bool, strategy: int, n_mol: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    L = _C.lib()
    dev = pos.device
    n = pos.shape[0]
    box_mode = 0 if box is None else (1 if box.dim() == 2 else 2)
    with torch.cuda.device(dev):
        nbytes = C.c_size_t(0)
        L.tmdnet_neighbor_workspace_bytes(n, n_mol, max_num_pairs, C.byref(nbytes))
        # scratch for the CSR build: one buffer per (device, stream) - two calls in flight on different streams must not share
        # it, and a buffer that is outgrown stays referenced by the launches already queued on ITS stream only (the caching
        # allocator frees stream-ordered on the allocating stream, which is the stream of those launches)
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < nbytes.value:
            ws = torch.empty(int(nbytes.value * 1.1) + 256, dtype=torch.uint8, device=dev)
            _ws_cache[key] = ws
        neighbors = torch.empty((2, max_num_pairs), dtype=torch.long, device=dev)
        f64 = pos.dtype == torch.float64  # the reference's operator is generic over the position dtype (warp_ops/neighbors.py:34-148)
        deltas = torch.empty((max_num_pairs, 3), dtype=pos.dtype, device=dev)
        dist = torch.empty((max_num_pairs,), dtype=pos.dtype, device=dev)
        num_pairs = torch.zeros(1, dtype=torch.int32, device=dev)
        p = pos.detach().contiguous()
        b = batch.contiguous()
        bx = None if box is None else box.detach().to(pos.dtype).contiguous()
        if f64:
            rc = L.tmdnet_neighbor_pairs_f64(_stream_ptr(dev), _ptr(ws), ws.numel(), n, n_mol, _ptr(p), _ptr(b), _ptr(bx), box_mode,
                                             float(cutoff_lower), float(cutoff_upper), max_num_pairs, int(loop), int(include_transpose),
                                             _ptr(neighbors), _ptr(deltas), _ptr(dist), _ptr(num_pairs))
        else:
            rc = L.tmdnet_neighbor_pairs(_stream_ptr(dev), _ptr(ws), ws.numel(), n, n_mol, _ptr(p), _ptr(b), _ptr(bx), box_mode,
                                         float(cutoff_lower), float(cutoff_upper), max_num_pairs, int(loop), int(include_transpose),
                                         int(strategy), _ptr(neighbors), _ptr(deltas), _ptr(dist), _ptr(num_pairs))
    if rc != _C.OK:
        raise RuntimeError(f"tmdnet_neighbor_pairs failed with code {rc}")
    return neighbors, deltas, dist, num_pairs


@neighbor_pairs.register_fake
def _(pos, batch, box, cutoff_lower, cutoff_upper, max_num_pairs, loop, include_transpose, strategy, n_mol):
    return (pos.new_empty((2, max_num_pairs), dtype=torch.long), pos.new_empty((max_num_pairs, 3)),
            pos.new_empty((max_num_pairs,)), pos.new_empty((1,), dtype=torch.int32))


@torch.library.custom_op("tmdnet::neighbor_grad", mutates_args=(), device_types="cuda")
def neighbor_grad(neighbors: Tensor, deltas: Tensor, distances: Tensor, grad_deltas: Optional[Tensor], grad_distances: Optional[Tensor],
                  n_atoms: int) -> Tensor:
    L = _C.lib()
    dev = deltas.device
    dt = deltas.dtype
    out = torch.empty((n_atoms, 3), dtype=dt, device=dev)
    gd = None if grad_deltas is None else grad_deltas.to(dt).contiguous()
    gw = None if grad_distances is None else grad_distances.to(dt).contiguous()
    fn = L.tmdnet_neighbor_grad_f64 if dt == torch.float64 else L.tmdnet_neighbor_grad
    with torch.cuda.device(dev):
        rc = fn(_stream_ptr(dev), _ptr(neighbors.contiguous()), _ptr(deltas.contiguous()), _ptr(distances.contiguous()),
                _ptr(gd), _ptr(gw), distances.shape[0], n_atoms, _ptr(out))
    if rc != _C.OK:
        raise RuntimeError(f"tmdnet_neighbor_grad failed with code {rc}")
    return out


@neighbor_grad.register_fake
def _(neighbors, deltas, distances, grad_deltas, grad_distances, n_atoms):
    return deltas.new_empty((n_atoms, 3))


# Second derivatives through the neighbour operator (reference tests/test_neighbors.py:272-315 runs gradgradcheck; the reference
# gets them for free because its backward, extensions/neighbor_utils.py:11-46, is written in differentiable torch ops).
# neighbor_grad is  out[i_p] += G_p, out[j_p] -= G_p  with  G_p = g_delta_p + delta_p g_dist_p / d_p  (padded and d = 0 entries
# contribute nothing).  With H the gradient arriving at `out` and h_p = H[i_p] - H[j_p]:
#     d/d g_delta_p = h_p ;  d/d g_dist_p = h_p . delta_p / d_p ;  d/d delta_p = h_p g_dist_p / d_p ;  d/d d_p = -(h_p . delta_p) g_dist_p / d_p^2
# Elementwise expressions on gathered rows, written in torch so that they are differentiable again; off every hot path.
def _ng_setup(ctx, inputs, output):
    neighbors, deltas, distances, grad_deltas, grad_distances, n_atoms = inputs
    ctx.have = (grad_deltas is not None, grad_distances is not None)
    ctx.save_for_backward(neighbors, deltas, distances, grad_distances)


def _ng_backward(ctx, H):
    neighbors, deltas, distances, grad_distances = ctx.saved_tensors
    have_gd, have_gw = ctx.have
    valid = (neighbors[0] >= 0) & (neighbors[1] >= 0) & (distances != 0)
    i = neighbors[0].clamp(min=0)
    j = neighbors[1].clamp(min=0)
    h = (H[i] - H[j]) * valid.unsqueeze(-1).to(H.dtype)
    safe_d = torch.where(valid, distances, torch.ones_like(distances))
    g_gdel = h if have_gd else None
    g_gdist = g_del = g_d = None
    if have_gw:
        hd = (h * deltas).sum(-1)
        g_gdist = hd / safe_d
        g_del = h * (grad_distances / safe_d).unsqueeze(-1)
        g_d = -hd * grad_distances / (safe_d * safe_d)
    return None, g_del, g_d, g_gdel, g_gdist, None


neighbor_grad.register_autograd(_ng_backward, setup_context=_ng_setup)


def _np_setup(ctx, inputs, output):
    neighbors, deltas, distances, num_pairs = output
    ctx.n_atoms = inputs[0].shape[0]
    ctx.save_for_backward(neighbors, deltas, distances)


def _np_backward(ctx, g_neighbors, g_deltas, g_distances, g_num_pairs):
    neighbors, deltas, distances = ctx.saved_tensors
    g_pos = torch.ops.tmdnet.neighbor_grad(neighbors, deltas, distances, g_deltas, g_distances, ctx.n_atoms)
    return g_pos, None, None, None, None, None, None, None, None, None


neighbor_pairs.register_autograd(_np_backward, setup_context=_np_setup)
