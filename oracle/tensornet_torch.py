"""TEST INFRASTRUCTURE ONLY -- pure-PyTorch CPU restatement of the TensorNet energy+force path.

Role: oracle for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py cpu_baseline leg).
It is never imported by the product package.  Forces come from ``torch.autograd`` here, which
makes this file an independent check of the hand-derived adjoints used by both
oracle/tensornet_c.c and the HIP kernels.

Pinned (tests/test_oracle.py) against
  * the reference's golden vector tests/expected.pkl["tensornet"]["Scalar"]
    (committed as tests/golden/expected_tensornet_scalar.pt), and
  * outputs of the unmodified reference imported from /root/reference
    (oracle/make_golden.py -> tests/golden/*.pt).

Every function cites the reference lines it restates (paths relative to /root/reference).
Functional style: the model is a flat ``state_dict`` with the reference's key names
 plus a small hyper-parameter dict.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as Fn

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# neighbor list: torchmdnet/models/utils.py:233-313 + extensions/warp_kernels/neighbors_brute.py
# ----------------------------------------------------------------------------------------------
def neighbor_pairs(pos: Tensor, batch: Tensor, cutoff_lower: float, cutoff_upper: float,
                   box: Optional[Tensor] = None, loop: bool = True, include_transpose: bool = True):
    """All pairs of one molecule with cutoff_lower <= d < cutoff_upper, both directions, plus one
    self edge per atom (d=0, delta=0) -- neighbors_brute.py:98-197, tests/test_neighbors.py:30-71.
    edge_vec = pos[edge_index[0]] - pos[edge_index[1]] (+ triclinic minimum image, z->y->x).
    Returned in a deterministic order (sorted by (i, j))."""
    n = pos.shape[0]
    same = batch[:, None] == batch[None, :]
    i, j = torch.nonzero(torch.tril(same, diagonal=-1), as_tuple=True)  # i > j
    delta = pos[i] - pos[j]
    if box is not None:
        b = box.to(pos.dtype)
        if b.dim() == 2:
            b = b.unsqueeze(0).expand(int(batch.max()) + 1, 3, 3)
        bb = b[batch[i]]
        delta = delta - torch.round(delta[:, 2] / bb[:, 2, 2])[:, None] * bb[:, 2]
        delta = delta - torch.round(delta[:, 1] / bb[:, 1, 1])[:, None] * bb[:, 1]
        delta = delta - torch.round(delta[:, 0] / bb[:, 0, 0])[:, None] * bb[:, 0]
    d2 = (delta * delta).sum(-1)
    keep = (d2 < cutoff_upper ** 2) & (d2 >= cutoff_lower ** 2)
    i, j, delta = i[keep], j[keep], delta[keep]
    ei, ej, dl = [i], [j], [delta]
    if include_transpose:
        ei.append(j), ej.append(i), dl.append(-delta)
    if loop:
        ar = torch.arange(n, device=pos.device)
        ei.append(ar), ej.append(ar), dl.append(torch.zeros(n, 3, dtype=pos.dtype, device=pos.device))
    ei, ej, dl = torch.cat(ei), torch.cat(ej), torch.cat(dl)
    order = torch.argsort(ei * n + ej)
    ei, ej, dl = ei[order], ej[order], dl[order]
    # distance with a safe gradient at d=0 (neighbor_utils.py:28-44 zeroes it)
    is_self = ei == ej
    safe = torch.where(is_self[:, None], torch.ones_like(dl), dl)
    dist = torch.where(is_self, torch.zeros_like(dl[:, 0]), safe.norm(dim=-1))
    return torch.stack([ei, ej]), dist, dl


# ----------------------------------------------------------------------------------------------
# radial functions: torchmdnet/models/utils.py:356-407 (ExpNormalSmearing), :500-528 (CosineCutoff)
# ----------------------------------------------------------------------------------------------
def cosine_cutoff(d: Tensor, lower: float, upper: float) -> Tensor:
    if lower > 0:  # utils.py:507-523
        c = 0.5 * (torch.cos(math.pi * (2 * (d - lower) / (upper - lower) + 1.0)) + 1.0)
        return c * (d < upper) * (d > lower)
    c = 0.5 * (torch.cos(d * math.pi / upper) + 1.0)  # utils.py:524-528
    return c * (d < upper)


def expnorm_rbf(d: Tensor, means: Tensor, betas: Tensor, lower: float, upper: float) -> Tensor:
    alpha = 5.0 / (upper - lower)  # utils.py:372
    d = d.unsqueeze(-1)
    return cosine_cutoff(d, 0.0, upper) * torch.exp(  # cutoff_fn has lower=0: utils.py:371
        -betas * (torch.exp(alpha * (-d + lower)) - means) ** 2
    )


def expnorm_initial_params(lower: float, upper: float, num_rbf: int, dtype=torch.float32):
    """utils.py:382-395."""
    start = torch.exp(torch.scalar_tensor(-upper + lower, dtype=dtype))
    means = torch.linspace(start, 1, num_rbf, dtype=dtype)
    betas = torch.tensor([(2 / num_rbf * (1 - start)) ** -2] * num_rbf, dtype=dtype)
    return means, betas


# ----------------------------------------------------------------------------------------------
# 3x3 tensor helpers on [N,3,3,F]: torchmdnet/models/tensornet.py:18-51, 84-146
# ----------------------------------------------------------------------------------------------
def _eye(t):
    return torch.eye(3, dtype=t.dtype, device=t.device)[None, :, :, None]


def decompose(X):  # tensornet.py:18-27
    A = 0.5 * (X - X.transpose(1, 2))
    I = X.diagonal(dim1=1, dim2=2).mean(-1)  # [N,F]
    S = 0.5 * (X + X.transpose(1, 2)) - I[:, None, None, :] * _eye(X)
    return I, A, S


def skew(v):  # tensornet.py:84-103 ; v: [N,3,F]
    z = torch.zeros_like(v[:, 0])
    return torch.stack(
        [z, -v[:, 2], v[:, 1], v[:, 2], z, -v[:, 0], -v[:, 1], v[:, 0], z], dim=1
    ).view(v.shape[0], 3, 3, v.shape[-1])


def tnorm(X):  # tensornet.py:144-146
    return (X ** 2).sum((1, 2))


def matmul33(A, B):  # per (atom, channel) 3x3 product, tensornet.py:42-51
    return torch.einsum("nabf,nbcf->nacf", A, B)


def lin(x, sd, key, bias=True):
    return Fn.linear(x, sd[key + ".weight"], sd[key + ".bias"] if bias else None)


def tensor_embedding(sd: Dict[str, Tensor], hp: dict, z: Tensor, edge_index: Tensor, d: Tensor, rhat: Tensor, phi: Tensor) -> Tensor:
    """TensorEmbedding.forward, tensornet.py:543-619 (shared by TensorNet and TensorNet2)."""
    P = "representation_model."
    F = hp["hidden_channels"]
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    n = z.shape[0]
    ei, ej = edge_index
    pos_dtype = d.dtype
    # ---- TensorEmbedding.forward, tensornet.py:543-619
    T = P + "tensor_embedding."
    Z = Fn.embedding(z, sd[T + "emb.weight"])
    Zij = lin(Z.index_select(0, edge_index.t().reshape(-1)).view(-1, 2 * F), sd, T + "emb2")  # :526-541
    C = cosine_cutoff(d, lo, up)
    W = (C[:, None] * Zij)[:, None, :] * torch.stack(
        [lin(phi, sd, T + "distance_proj1"), lin(phi, sd, T + "distance_proj2"), lin(phi, sd, T + "distance_proj3")],
        dim=1,
    )  # [E,3,F]  :558-567
    # tensornet_embedding_message_passing, :405-445
    I0 = torch.zeros(n, F, dtype=pos_dtype, device=W.device).index_add(0, ei, W[:, 0])
    v0 = torch.zeros(n, 3, F, dtype=pos_dtype, device=W.device).index_add(0, ei, W[:, 1, None, :] * rhat[:, :, None])
    outer = rhat[:, :, None] * rhat[:, None, :]
    T0 = torch.zeros(n, 3, 3, F, dtype=pos_dtype, device=W.device).index_add(0, ei, W[:, 2, None, None, :] * outer[..., None])
    A0 = skew(v0)
    S0 = 0.5 * (T0 + T0.transpose(1, 2)) - T0.diagonal(dim1=1, dim2=2).mean(-1)[:, None, None, :] * _eye(T0)  # :133-141
    X = I0[:, None, None, :] * _eye(T0) + A0 + S0
    norm = Fn.layer_norm(tnorm(X), (F,), sd[T + "init_norm.weight"], sd[T + "init_norm.bias"])  # :589
    norm = Fn.silu(lin(norm, sd, T + "linears_scalar.0"))
    norm = Fn.silu(lin(norm, sd, T + "linears_scalar.1")).reshape(n, 3, F)  # :590-593
    I1 = lin(I0, sd, T + "linears_tensor.0", bias=False) * norm[:, 0]
    A1 = lin(A0, sd, T + "linears_tensor.1", bias=False) * norm[:, 1, None, None, :]
    S1 = lin(S0, sd, T + "linears_tensor.2", bias=False) * norm[:, 2, None, None, :]
    X = I1[:, None, None, :] * _eye(T0) + A1 + S1  # :617
    return X


# ----------------------------------------------------------------------------------------------
# TensorNet representation: torchmdnet/models/tensornet.py:308-402 (OPT=False branch)
# ----------------------------------------------------------------------------------------------
def tensornet_representation(sd: Dict[str, Tensor], hp: dict, z: Tensor, pos: Tensor, batch: Tensor,
                             box: Optional[Tensor] = None, q: Optional[Tensor] = None,
                             return_intermediates: bool = False):
    P = "representation_model."
    F = hp["hidden_channels"]
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    n = z.shape[0]
    edge_index, d, vec = neighbor_pairs(pos, batch, lo, up, box)
    ei, ej = edge_index
    qa = torch.zeros(n, dtype=pos.dtype, device=pos.device) if q is None else q.to(pos.dtype)[batch]  # :341-344
    phi = expnorm_rbf(d, sd[P + "distance_expansion.means"], sd[P + "distance_expansion.betas"], lo, up)  # :362
    is_self = ei == ej
    rhat = vec / torch.where(is_self, torch.ones_like(d), d)[:, None]  # :363-366

    X = tensor_embedding(sd, hp, z, edge_index, d, rhat, phi)
    C = cosine_cutoff(d, lo, up)
    inter = {"edge_index": edge_index, "d": d, "rhat": rhat, "phi": phi, "X_embed": X}

    # ---- Interaction.forward x L, tensornet.py:729-814
    kappa = (1 + 0.1 * qa)[:, None, None, None]
    for l in range(hp["num_layers"]):
        Lp = P + f"layers.{l}."
        w = phi
        for k in range(3):
            w = Fn.silu(lin(w, sd, Lp + f"linears_scalar.{k}"))  # :739-740
        w = (w * C[:, None]).reshape(-1, 3, F)  # :741-743
        X = X / (tnorm(X) + 1)[:, None, None, :]  # :745
        I, A, S = decompose(X)
        I = lin(I, sd, Lp + "linears_tensor.0", bias=False)
        A = lin(A, sd, Lp + "linears_tensor.1", bias=False)
        S = lin(S, sd, Lp + "linears_tensor.2", bias=False)
        Y = I[:, None, None, :] * _eye(X) + A + S  # :755
        # message passing :622-679, same op sequence as the reference (this function is also bench.py's CPU baseline):
        # the skew part travels as a vector (skewtensor_to_vector :106-123), gathers are index_select, sums index_add
        Af = A.flatten(1, 2)
        Av = 0.5 * torch.stack((Af[:, 7] - Af[:, 5], Af[:, 2] - Af[:, 6], Af[:, 3] - Af[:, 1]), dim=1)
        Im = torch.zeros_like(I).index_add(0, ei, w[:, 0] * I.index_select(0, ej))
        Avm = torch.zeros_like(Av).index_add(0, ei, w[:, 1, None, :] * Av.index_select(0, ej))
        Sm = torch.zeros_like(S).index_add(0, ei, w[:, 2, None, None, :] * S.index_select(0, ej))
        M = Im[:, None, None, :] * _eye(X) + skew(Avm) + Sm
        if hp.get("equivariance_invariance_group", "O(3)") == "O(3)":
            Cm = kappa * (matmul33(M, Y) + matmul33(Y, M))  # :788-790
        else:
            Cm = 2 * matmul33(Y, M)  # :791-793
        I, A, S = decompose(Cm)
        np1 = tnorm(Cm) + 1  # :795
        I = lin(I / np1, sd, Lp + "linears_tensor.3", bias=False)
        A = lin(A / np1[:, None, None, :], sd, Lp + "linears_tensor.4", bias=False)
        S = lin(S / np1[:, None, None, :], sd, Lp + "linears_tensor.5", bias=False)
        dX = I[:, None, None, :] * _eye(X) + A + S
        X = X + dX + kappa * matmul33(dX, dX)  # :812
        inter[f"X_layer{l}"] = X

    # ---- readout in TensorNet.forward, tensornet.py:384-398
    I, A, S = decompose(X)
    x = torch.cat([3 * I ** 2, tnorm(A), tnorm(S)], dim=-1)
    x = Fn.layer_norm(x, (3 * F,), sd[P + "out_norm.weight"], sd[P + "out_norm.bias"])
    x = Fn.silu(lin(x, sd, P + "linear"))
    if return_intermediates:
        return x, inter
    return x


# ----------------------------------------------------------------------------------------------
# TorchMD_Net.forward with the Scalar head: torchmdnet/models/model.py:530-631,
# output_modules.py:43-117, models/utils.py:552-580, priors/atomref.py:93-96
# ----------------------------------------------------------------------------------------------
def energy(sd, hp, z, pos, batch, box=None, q=None, num_systems=None, atomref=None, atom_weights=None):
    """``atom_weights`` is NOT a reference argument: it restates include/tmdnet_amd.h's tmdnet_set_atom_weights (E = sum_i w_i e_i)
    so that the domain-decomposition tests have a checker for the weighted sum."""
    x = tensornet_representation(sd, hp, z, pos, batch, box, q)
    O = "output_model.output_network.layers."
    h = Fn.silu(lin(x, sd, O + "0"))
    k = 2
    while (O + f"{k + 2}.weight") in sd:  # optional hidden layers (output_mlp_num_layers)
        h = Fn.silu(lin(h, sd, O + f"{k}"))
        k += 2
    e = lin(h, sd, O + f"{k}")  # [N,1]
    e = e * sd.get("std", torch.ones((), dtype=e.dtype))  # model.py:594-595
    if atomref is not None:  # priors/atomref.py:93-96
        e = e + atomref[z]
    if atom_weights is not None:
        e = e * atom_weights.to(e.dtype).reshape(-1, 1)
    nmol = int(batch.max()) + 1 if num_systems is None else num_systems
    y = torch.zeros(nmol, 1, dtype=e.dtype, device=e.device).index_add(0, batch, e)  # output_modules.py:43-73
    return y + sd.get("mean", torch.zeros((), dtype=e.dtype, device=e.device))  # model.py:606-607


def energy_and_forces(sd, hp, z, pos, batch, box=None, q=None, num_systems=None, atomref=None, atom_weights=None):
    """model.py:584-628: y, -dy with dy = grad(sum y, pos)."""
    pos = pos.detach().clone().requires_grad_(True)
    y = energy(sd, hp, z, pos, batch, box, q, num_systems, atomref, atom_weights)
    (dy,) = torch.autograd.grad([y], [pos], grad_outputs=[torch.ones_like(y)])
    return y.detach(), -dy


def hparams_from_args(args: dict) -> dict:
    """Subset of the create_model argument dict the path depends on (model.py:35-60,96-105)."""
    return dict(
        hidden_channels=int(args["embedding_dimension"]),
        num_layers=int(args["num_layers"]),
        num_rbf=int(args["num_rbf"]),
        cutoff_lower=float(args["cutoff_lower"]),
        cutoff_upper=float(args["cutoff_upper"]),
        max_z=int(args["max_z"]),
        max_num_neighbors=int(args["max_num_neighbors"]),
        equivariance_invariance_group=args.get("equivariance_invariance_group", "O(3)"),
    )


def cast_state_dict(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
