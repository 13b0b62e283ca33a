"""TEST INFRASTRUCTURE ONLY -- torch restatement of the reference's TensorNet2 + ScalarPlusWeightedCoulomb path
(torchmdnet/models/tensornet2.py:49-626, torchmdnet/models/output_modules.py:323-609, model.py:530-631), pure-PyTorch branch
(OPT = False), autograd forces.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

TensorNet2 = TensorNet's embedding and interaction layers + AIMNet2-style charge equilibration: after the embedding and after
every layer a ChargePredict head turns the invariants [I, |A|^2, |S|^2] into q_dim partial-charge channels per atom and
equilibrates them per molecule to the total charge; the next layer's edge MLP takes [phi(d), c_i, c_j] (so its weights are per
DIRECTED edge); the output head adds a damped pair Coulomb energy of all charge channels (weighted mean) to the scalar head.
Pinned: oracle/make_golden_tn2.py writes fixtures from the UNMODIFIED reference; tests/test_oracle.py compares (fp64 1e-12).
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as Fn
from torch import Tensor

from oracle import tensornet_torch as TT

_HARTREE = 27.211386024367243
_BOHR = 0.5291772105638411
COULOMB_FACTOR = 0.5 * _HARTREE * _BOHR  # output_modules.py:399-403


def exp_cutoff(d, rc: float):  # output_modules.py:323-332
    return torch.exp(-1.0 / (1.0 - (d / rc).clamp(0, 1.0 - 1e-6).pow(2))) / 0.36787944117144233


def charge_predict(sd, pre, X, batch, Qatom, nmol, q_dim):  # tensornet2.py:49-157
    I, A, S = TT.decompose(X)
    x = torch.cat((I, TT.tnorm(A), TT.tnorm(S)), dim=-1)  # NB: I itself, not 3 I^2 (:141)
    F3 = x.shape[-1]
    x = Fn.layer_norm(x, (F3,), sd[pre + "q_norm.weight"], sd[pre + "q_norm.bias"])
    h = Fn.silu(TT.lin(x, sd, pre + "q_mlp.layers.0"))
    k = 2
    while (pre + f"q_mlp.layers.{k + 2}.weight") in sd:
        h = Fn.silu(TT.lin(h, sd, pre + f"q_mlp.layers.{k}"))
        k += 2
    out = TT.lin(h, sd, pre + f"q_mlp.layers.{k}")
    c, f = out[:, :q_dim], out[:, q_dim:]
    fu = f ** 2  # qeq, :99-138
    Fu = torch.zeros(nmol, q_dim, dtype=X.dtype).index_add(0, batch, fu) + 1.0e-6
    Qu = torch.zeros(nmol, q_dim, dtype=X.dtype).index_add(0, batch, c)
    dQ = Qatom.unsqueeze(-1) - Qu[batch]
    return c + fu / Fu[batch] * dQ


def tn2_representation(sd: Dict[str, Tensor], hp: dict, z, pos, batch, box=None, q=None, nmol=None, inter=None):
    P = "representation_model."
    F, qd = hp["hidden_channels"], hp["q_dim"]
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    n = z.shape[0]
    nmol = int(batch.max()) + 1 if nmol is None else nmol
    edge_index, d, vec = TT.neighbor_pairs(pos, batch, lo, up, box)
    ei, ej = edge_index
    Qatom = torch.zeros(n, dtype=pos.dtype) if q is None else q.to(pos.dtype)[batch]  # :411-414
    phi = TT.expnorm_rbf(d, sd[P + "distance_expansion.means"], sd[P + "distance_expansion.betas"], lo, up)
    is_self = ei == ej
    rhat = vec / torch.where(is_self, torch.ones_like(d), d)[:, None]
    C = TT.cosine_cutoff(d, lo, up)
    # TensorEmbedding: identical to TensorNet's (tensornet.py:543-619)
    X = TT.tensor_embedding(sd, hp, z, edge_index, d, rhat, phi)
    charges = [charge_predict(sd, P + "charge_predict_0.", X, batch, Qatom, nmol, qd)]
    if inter is not None:
        inter["X_embed"] = X
        inter["charges_0"] = charges[0]
    for l in range(hp["num_layers"]):
        Lp = P + f"layers.{l}."
        c = charges[-1]
        w = torch.cat([phi, c.index_select(0, ei), c.index_select(0, ej)], dim=-1)  # :548-566
        for k in range(3):
            w = Fn.silu(TT.lin(w, sd, Lp + f"linears_scalar.{k}"))
        w = (w * C[:, None]).reshape(-1, 3, F)
        X = X / (TT.tnorm(X) + 1)[:, None, None, :]
        I, A, S = TT.decompose(X)
        I = TT.lin(I, sd, Lp + "linears_tensor.0", bias=False)
        A = TT.lin(A, sd, Lp + "linears_tensor.1", bias=False)
        S = TT.lin(S, sd, Lp + "linears_tensor.2", bias=False)
        Y = I[:, None, None, :] * TT._eye(X) + A + S
        Af = A.flatten(1, 2)
        Av = 0.5 * torch.stack((Af[:, 7] - Af[:, 5], Af[:, 2] - Af[:, 6], Af[:, 3] - Af[:, 1]), dim=1)
        Im = torch.zeros_like(I).index_add(0, ei, w[:, 0] * I.index_select(0, ej))
        Avm = torch.zeros_like(Av).index_add(0, ei, w[:, 1, None, :] * Av.index_select(0, ej))
        Sm = torch.zeros_like(S).index_add(0, ei, w[:, 2, None, None, :] * S.index_select(0, ej))
        M = Im[:, None, None, :] * TT._eye(X) + TT.skew(Avm) + Sm
        if hp.get("equivariance_invariance_group", "O(3)") == "O(3)":
            Cm = TT.matmul33(M, Y) + TT.matmul33(Y, M)  # tensor_matmul_o3, :608-610
        else:
            Cm = 2 * TT.matmul33(Y, M)
        I, A, S = TT.decompose(Cm)
        np1 = TT.tnorm(Cm) + 1
        I = TT.lin(I / np1, sd, Lp + "linears_tensor.3", bias=False)
        A = TT.lin(A / np1[:, None, None, :], sd, Lp + "linears_tensor.4", bias=False)
        S = TT.lin(S / np1[:, None, None, :], sd, Lp + "linears_tensor.5", bias=False)
        dX = I[:, None, None, :] * TT._eye(X) + A + S
        X = X + dX + TT.matmul33(dX, dX)  # :624 (no charge factor here)
        charges.append(charge_predict(sd, P + f"charge_predicts.{l}.", X, batch, Qatom, nmol, qd))
        if inter is not None:
            inter[f"X_layer{l}"] = X
            inter[f"charges_{l + 1}"] = charges[-1]
    I, A, S = TT.decompose(X)
    x = torch.cat([3 * I ** 2, TT.tnorm(A), TT.tnorm(S)], dim=-1)
    x = Fn.layer_norm(x, (3 * F,), sd[P + "out_norm.weight"], sd[P + "out_norm.bias"])
    x = Fn.silu(TT.lin(x, sd, P + "linear"))
    return x, torch.cat(charges, dim=-1)


def coulomb_per_atom(hp, charges, pos, batch, qweights, box=None):  # output_modules.py:440-606
    n = pos.shape[0]
    cut = hp.get("coulomb_cutoff")
    if cut is None:
        if box is not None:
            raise ValueError("PBC is not supported with coulomb_cutoff = None")
        i, j = torch.triu_indices(n, n, 1).unbind(0)
        m = batch[i] == batch[j]
        i, j = i[m], j[m]
        d = torch.linalg.norm(pos[i] - pos[j], dim=-1)
        g = 1.0 / d
    else:
        ei, d, _ = TT.neighbor_pairs(pos, batch, 0.0, float(cut), box, loop=False, include_transpose=False)
        i, j = ei
        eps = float(hp.get("coulomb_epsilon_solvent", 78.3))
        k_rf = (1.0 / cut ** 3) * (eps - 1.0) / (2.0 * eps + 1.0)
        c_rf = (1.0 / cut) * (3.0 * eps) / (2.0 * eps + 1.0)
        g = 1.0 / d + k_rf * d ** 2 - c_rf
    fc = 1.0 - exp_cutoff(d, 4.6)
    e = COULOMB_FACTOR * (fc * g).unsqueeze(-1) * charges[i] * charges[j]
    e = torch.sum(e * qweights.unsqueeze(0), dim=-1) / torch.sum(qweights)
    out = torch.zeros(n, dtype=pos.dtype)
    return out.index_add(0, i, e).index_add(0, j, e)


def energy(sd, hp, z, pos, batch, box=None, q=None, num_systems=None, atomref=None, inter=None):
    nmol = int(batch.max()) + 1 if num_systems is None else num_systems
    x, charges = tn2_representation(sd, hp, z, pos, batch, box, q, nmol, inter)
    O = "output_model.output_network.layers."
    h = Fn.silu(TT.lin(x, sd, O + "0"))
    k = 2
    while (O + f"{k + 2}.weight") in sd:
        h = Fn.silu(TT.lin(h, sd, O + f"{k}"))
        k += 2
    e = TT.lin(h, sd, O + f"{k}")
    e = e + coulomb_per_atom(hp, charges, pos, batch, sd["output_model.qweights"].to(pos.dtype), box).unsqueeze(-1)
    e = e * sd.get("std", torch.ones((), dtype=e.dtype))
    if atomref is not None:
        e = e + atomref[z]
    y = torch.zeros(nmol, 1, dtype=e.dtype).index_add(0, batch, e)
    if inter is not None:
        inter["charges"] = charges
    return y + sd.get("mean", torch.zeros((), dtype=e.dtype))


def energy_and_forces(sd, hp, z, pos, batch, box=None, q=None, num_systems=None, atomref=None):
    pos = pos.detach().clone().requires_grad_(True)
    y = energy(sd, hp, z, pos, batch, box, q, num_systems, atomref)
    (dy,) = torch.autograd.grad([y], [pos], grad_outputs=[torch.ones_like(y)])
    return y.detach(), -dy


def hparams_from_args(args: dict) -> dict:
    hp = TT.hparams_from_args(args)
    hp.update(q_dim=int(args.get("q_dim", 0)), coulomb_cutoff=args.get("coulomb_cutoff"),
              coulomb_epsilon_solvent=args.get("coulomb_epsilon_solvent", 78.3))
    return hp
