"""TEST INFRASTRUCTURE ONLY -- pure-PyTorch CPU restatement of the Equivariant Transformer energy+force path
(BASELINE configs[3] = examples/ET-SPICE.yaml).

Role: oracle for the HIP kernels of the ET path (tests/, never imported by the product package).  Forces come
from ``torch.autograd`` here, an independent check of the hand-derived adjoints in the HIP kernels.

Pinned (tests/test_oracle.py) against
  * the reference's golden vector tests/expected.pkl["equivariant-transformer"]["Scalar"]
    (committed as tests/golden/expected_et_scalar.pt), and
  * outputs of the unmodified reference imported from /root/reference (oracle/make_golden_et.py ->
    tests/golden/et_*.pt).

Every function cites the reference lines it restates (paths relative to /root/reference).  Functional style: the
model is a flat ``state_dict`` with the reference's key names plus a small hyper-parameter dict.
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as Fn

from oracle.tensornet_torch import cosine_cutoff, expnorm_rbf, neighbor_pairs

Tensor = torch.Tensor


def lin(x, sd, key, bias=True):
    return Fn.linear(x, sd[key + ".weight"], sd[key + ".bias"] if bias and (key + ".bias") in sd else None)


def et_representation(sd: Dict[str, Tensor], hp: dict, z: Tensor, pos: Tensor, batch: Tensor, box: Optional[Tensor] = None,
                      return_intermediates: bool = False):
    """TorchMD_ET.forward (torchmdnet/models/torchmd_et.py:188-221) -> x [N,F], vec [N,3,F]."""
    R = "representation_model."
    F, H = hp["hidden_channels"], hp["num_heads"]
    hd = F // H
    lo, up = hp["cutoff_lower"], hp["cutoff_upper"]
    inter = {}
    x = sd[R + "embedding.weight"][z]  # :195
    # :197 OptimizedDistance(loop=True): self loops, both directions; edge_vec = pos[ei0] - pos[ei1]
    edge_index, d, vec_ij = neighbor_pairs(pos, batch, lo, up, box=box, loop=True)
    phi = expnorm_rbf(d, sd[R + "distance_expansion.means"], sd[R + "distance_expansion.betas"], lo, up)  # :205
    mask = edge_index[0] != edge_index[1]
    rhat = torch.where(mask[:, None], vec_ij / torch.where(mask, d, torch.ones_like(d))[:, None], torch.zeros_like(vec_ij))  # :206-207
    C = cosine_cutoff(d, lo, up)

    if hp["neighbor_embedding"]:  # models/utils.py:83-117 (self loops removed)
        Np = R + "neighbor_embedding."
        W = lin(phi[mask], sd, Np + "distance_proj") * C[mask][:, None]
        msg = W * sd[Np + "embedding.weight"][z][edge_index[1][mask]]
        xn = torch.zeros_like(x).index_add(0, edge_index[0][mask], msg)
        x = lin(torch.cat([x, xn], dim=1), sd, Np + "combine")
    inter["x_embed"] = x
    vec = torch.zeros(x.shape[0], 3, F, dtype=x.dtype)

    tgt, src = edge_index[1], edge_index[0]  # :368-373: messages flow source = ei0 -> target = ei1
    for l in range(hp["num_layers"]):  # EquivariantMultiHeadAttention.forward, :315-353
        Lp = R + f"attention_layers.{l}."
        xt = Fn.layer_norm(x, (F,), sd[Lp + "layernorm.weight"], sd[Lp + "layernorm.bias"])
        q = lin(xt, sd, Lp + "q_proj").reshape(-1, H, hd)
        k = lin(xt, sd, Lp + "k_proj").reshape(-1, H, hd)
        v = lin(xt, sd, Lp + "v_proj").reshape(-1, H, 3 * hd)
        vec1, vec2, vec3 = torch.split(lin(vec, sd, Lp + "vec_proj", bias=False), F, dim=-1)
        vec_dot = (vec1 * vec2).sum(dim=1)
        dk = Fn.silu(lin(phi, sd, Lp + "dk_proj")).reshape(-1, H, hd) if (Lp + "dk_proj.weight") in sd else None
        dv = Fn.silu(lin(phi, sd, Lp + "dv_proj")).reshape(-1, H, 3 * hd) if (Lp + "dv_proj.weight") in sd else None
        # message (:376-415)
        qi, kj, vj, vecj = q[tgt], k[src], v[src], vec.reshape(-1, 3, H, hd)[src]
        attn = (qi * kj).sum(-1) if dk is None else (qi * kj * dk).sum(-1)
        attn = Fn.silu(attn)
        if hp["vector_cutoff"]:
            vj = vj * C[:, None, None]
        else:
            attn = attn * C[:, None]
        if dv is not None:
            vj = vj * dv
        xm, v1, v2 = torch.split(vj, hd, dim=2)
        xm = xm * attn.unsqueeze(2)
        vm = vecj * v1.unsqueeze(1) + v2.unsqueeze(1) * rhat.unsqueeze(2).unsqueeze(3)
        # aggregate into the target (:417-426)
        xagg = torch.zeros(x.shape[0], H, hd, dtype=x.dtype).index_add(0, tgt, xm).reshape(-1, F)
        vagg = torch.zeros(x.shape[0], 3, H, hd, dtype=x.dtype).index_add(0, tgt, vm).reshape(-1, 3, F)
        o1, o2, o3 = torch.split(lin(xagg, sd, Lp + "o_proj"), F, dim=1)  # :347-353
        dx = vec_dot * o2 + o3
        dvec = vec3 * o1.unsqueeze(1) + vagg
        x = x + dx  # :215-217
        vec = vec + dvec
        inter[f"x_layer{l}"] = x
        inter[f"vec_layer{l}"] = vec
    x = Fn.layer_norm(x, (F,), sd[R + "out_norm.weight"], sd[R + "out_norm.bias"])  # :219
    if return_intermediates:
        return x, vec, inter
    return x, vec


def gated_block(sd, pre, x, v, scalar_act):
    """GatedEquivariantBlock.forward (torchmdnet/models/utils.py:626-655)."""
    v1b = Fn.linear(v, sd[pre + "vec1_proj.weight"])
    nz = (v1b != 0).reshape(v1b.shape[0], -1).any(dim=1)
    # zero rows are masked out of the norm (no NaN gradient): utils.py:630-644
    safe = torch.where(nz[:, None, None], v1b, torch.ones_like(v1b))
    vec1 = torch.where(nz[:, None], torch.norm(safe, dim=-2), torch.zeros_like(v1b[:, 0]))
    vec2 = Fn.linear(v, sd[pre + "vec2_proj.weight"])
    h = torch.cat([x, vec1], dim=-1)
    h = Fn.silu(lin(h, sd, pre + "update_net.layers.0"))
    h = lin(h, sd, pre + "update_net.layers.2")
    oc = h.shape[-1] // 2
    xo, g = torch.split(h, oc, dim=-1)
    vo = g.unsqueeze(1) * vec2
    if scalar_act:
        xo = Fn.silu(xo)
    return xo, vo


def energy(sd, hp, z, pos, batch, box=None, num_systems=None, atomref=None):
    """TorchMD_Net.forward with EquivariantScalar (model.py:530-631, output_modules.py:120-163)."""
    x, v = et_representation(sd, hp, z, pos, batch, box)
    O = "output_model.output_network."
    x, v = gated_block(sd, O + "0.", x, v, True)
    x, v = gated_block(sd, O + "1.", x, v, False)
    e = x + v.sum() * 0  # output_modules.py:163
    e = e * sd.get("std", torch.ones((), dtype=e.dtype))
    if atomref is not None:
        e = e + atomref[z]
    nmol = int(batch.max()) + 1 if num_systems is None else num_systems
    y = torch.zeros(nmol, 1, dtype=e.dtype).index_add(0, batch, e)
    return y + sd.get("mean", torch.zeros((), dtype=e.dtype))


def energy_and_forces(sd, hp, z, pos, batch, box=None, num_systems=None, atomref=None):
    pos = pos.detach().clone().requires_grad_(True)
    y = energy(sd, hp, z, pos, batch, box, num_systems, atomref)
    (dy,) = torch.autograd.grad([y], [pos], grad_outputs=[torch.ones_like(y)])
    return y.detach(), -dy


def hparams_from_args(args: dict) -> dict:
    """Subset of the create_model argument dict the ET path depends on (model.py:62-81)."""
    assert args.get("activation", "silu") == "silu" and args.get("attn_activation", "silu") == "silu"
    return dict(
        hidden_channels=int(args["embedding_dimension"]),
        num_layers=int(args["num_layers"]),
        num_rbf=int(args["num_rbf"]),
        num_heads=int(args["num_heads"]),
        cutoff_lower=float(args["cutoff_lower"]),
        cutoff_upper=float(args["cutoff_upper"]),
        max_z=int(args["max_z"]),
        max_num_neighbors=int(args["max_num_neighbors"]),
        neighbor_embedding=bool(args.get("neighbor_embedding", True)),
        vector_cutoff=bool(args.get("vector_cutoff", False)),
        distance_influence=args.get("distance_influence", "both"),
    )
