"""TEST INFRASTRUCTURE ONLY -- kernel-level specification of the Equivariant Transformer E+F path with a
HAND-DERIVED reverse pass (no autograd), in the data layout and operation order the HIP kernels use.

Forward quantities per undirected pair (functions of d only: phi, C, dk, dv, the neighbour-embedding filter Wn) carry
their d/dd tangents forward, exactly as the TensorNet path does; the reverse pass reduces every per-pair gradient to
the scalar g_d[p] and the vector g_rhat[p] and never forms g_dk / g_dv.  Checked in fp64 against autograd of
oracle/et_torch.py (tests/test_oracle.py), which is itself pinned to the unmodified reference.

Index convention (reference torchmd_et.py:368-373): an edge carries a message from source s = edge_index[0] to
target t = edge_index[1]; rhat(t<-s) = (pos_s - pos_t) / d.  `cache=True` returns the intermediates the GPU tests
compare against.
"""
import math

import torch

from oracle.tensornet_torch import neighbor_pairs


def silu(x):
    return x * torch.sigmoid(x)


def dsilu(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def radial(d, means, betas, lo, up):
    """phi[K], dphi/dd, C, dC/dd (models/utils.py:402-407, 506-528; lower cutoff 0 as in the configs of the path)."""
    assert lo == 0.0
    alpha = 5.0 / (up - lo)
    inside = (d < up).to(d.dtype)
    c0 = 0.5 * (torch.cos(d * math.pi / up) + 1.0) * inside
    dc0 = -0.5 * math.pi / up * torch.sin(d * math.pi / up) * inside
    ex = torch.exp(alpha * (-d + lo))[:, None]
    g = torch.exp(-betas * (ex - means) ** 2)
    dg = g * (-2 * betas * (ex - means)) * (-alpha * ex)
    return c0[:, None] * g, dc0[:, None] * g + c0[:, None] * dg, c0, dc0


def layer_norm(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    xh = (x - mu) * rstd
    return xh * w + b, xh, rstd


def layer_norm_bwd(g, xh, rstd, w):
    gw = g * w
    return rstd * (gw - gw.mean(-1, keepdim=True) - xh * (gw * xh).mean(-1, keepdim=True))


def energy_forces(sd, hp, z, pos, batch, box=None, atomref=None, want_cache=False):
    R = "representation_model."
    F, H, L = hp["hidden_channels"], hp["num_heads"], hp["num_layers"]
    hd = F // H
    lo, up = hp["cutoff_lower"], hp["cutoff_upper"]
    vc = hp["vector_cutoff"]
    N = z.shape[0]
    dt = pos.dtype
    ei, d, delta = neighbor_pairs(pos, batch, lo, up, box=box, loop=True)
    src, tgt = ei[0], ei[1]
    E = src.shape[0]
    nself = src != tgt
    rhat = torch.where(nself[:, None], delta / torch.where(nself, d, torch.ones_like(d))[:, None], torch.zeros_like(delta))
    phi, dphi, C, dC = radial(d, sd[R + "distance_expansion.means"], sd[R + "distance_expansion.betas"], lo, up)

    def W(k):
        return sd[k + ".weight"]

    def B(k):
        return sd[k + ".bias"]

    # ---------------- forward
    x0 = W(R + "embedding")[z]
    ne = hp["neighbor_embedding"]
    if ne:
        Np = R + "neighbor_embedding."
        e_n = phi @ W(Np + "distance_proj").T + B(Np + "distance_proj")
        de_n = dphi @ W(Np + "distance_proj").T
        Wn, dWn = e_n * C[:, None], de_n * C[:, None] + e_n * dC[:, None]
        embN = W(Np + "embedding")[z]
        msg = torch.where(nself[:, None], Wn * embN[tgt], torch.zeros_like(Wn))  # reference: target of THIS sum is ei0
        xn = torch.zeros(N, F, dtype=dt).index_add(0, src, msg)
        Wc = W(Np + "combine")
        x = x0 @ Wc[:, :F].T + xn @ Wc[:, F:].T + B(Np + "combine")
    else:
        x = x0
    vec = torch.zeros(N, 3, F, dtype=dt)
    hix = torch.arange(F) // hd  # head of channel c
    cache = {"x_embed": x}
    saved = []
    for l in range(L):
        Lp = R + f"attention_layers.{l}."
        xt, xh, rstd = layer_norm(x, W(Lp + "layernorm"), B(Lp + "layernorm"))
        q = xt @ W(Lp + "q_proj").T + B(Lp + "q_proj")
        k = xt @ W(Lp + "k_proj").T + B(Lp + "k_proj")
        v = xt @ W(Lp + "v_proj").T + B(Lp + "v_proj")  # [N,3F], layout [H][3][hd]
        vp = vec @ W(Lp + "vec_proj").T  # [N,3,3F]
        vec1, vec2, vec3 = vp[..., :F], vp[..., F:2 * F], vp[..., 2 * F:]
        vdot = (vec1 * vec2).sum(1)
        has_dk, has_dv = (Lp + "dk_proj.weight") in sd, (Lp + "dv_proj.weight") in sd
        if has_dk:
            ek = phi @ W(Lp + "dk_proj").T + B(Lp + "dk_proj")
            dk, ddk = silu(ek), dsilu(ek) * (dphi @ W(Lp + "dk_proj").T)
        else:
            dk, ddk = torch.ones(E, F, dtype=dt), torch.zeros(E, F, dtype=dt)
        if has_dv:
            ev = phi @ W(Lp + "dv_proj").T + B(Lp + "dv_proj")
            dv, ddv = silu(ev), dsilu(ev) * (dphi @ W(Lp + "dv_proj").T)
        else:
            dv, ddv = torch.ones(E, 3 * F, dtype=dt), torch.zeros(E, 3 * F, dtype=dt)
        # channel c = h*hd + c' of the three value thirds lives at v[:, h*3hd + {0,hd,2hd} + c']
        cidx = torch.arange(F)
        ix = (cidx // hd) * 3 * hd + cidx % hd
        vx, v1, v2 = v[:, ix], v[:, ix + hd], v[:, ix + 2 * hd]
        dvx, dv1, dv2 = dv[:, ix], dv[:, ix + hd], dv[:, ix + 2 * hd]
        ddvx, ddv1, ddv2 = ddv[:, ix], ddv[:, ix + hd], ddv[:, ix + 2 * hd]
        cv = C if vc else torch.ones_like(C)
        ca = torch.ones_like(C) if vc else C
        a = torch.zeros(E, H, dtype=dt).index_add(1, hix, q[tgt] * k[src] * dk)  # [E,H]
        A = silu(a) * ca[:, None]
        Ac = A[:, hix]  # per channel
        sx, s1, s2 = vx[src] * cv[:, None] * dvx, v1[src] * cv[:, None] * dv1, v2[src] * cv[:, None] * dv2
        xagg = torch.zeros(N, F, dtype=dt).index_add(0, tgt, sx * Ac)
        vm = vec[src] * s1[:, None, :] + s2[:, None, :] * rhat[:, :, None]
        vagg = torch.zeros(N, 3, F, dtype=dt).index_add(0, tgt, vm)
        o = xagg @ W(Lp + "o_proj").T + B(Lp + "o_proj")
        o1, o2, o3 = o[:, :F], o[:, F:2 * F], o[:, 2 * F:]
        dx = vdot * o2 + o3
        dvec = vec3 * o1[:, None, :] + vagg
        saved.append(dict(x=x, vec=vec, xh=xh, rstd=rstd, q=q, k=k, vx=vx, v1=v1, v2=v2, vec1=vec1, vec2=vec2, vec3=vec3, vdot=vdot,
                          dk=dk, ddk=ddk, dvx=dvx, dv1=dv1, dv2=dv2, ddvx=ddvx, ddv1=ddv1, ddv2=ddv2, a=a, cv=cv, ca=ca,
                          o1=o1, o2=o2, ix=ix))
        x = x + dx
        vec = vec + dvec
        cache[f"x_layer{l}"], cache[f"vec_layer{l}"] = x, vec
    xf, xfh, rstdf = layer_norm(x, W(R + "out_norm"), B(R + "out_norm"))
    cache["x_out"] = xf

    # head (models/utils.py:626-655, output_modules.py:120-163)
    O0, O1 = "output_model.output_network.0.", "output_model.output_network.1."
    u1 = vec @ W(O0 + "vec1_proj").T
    n1 = torch.sqrt((u1 * u1).sum(1))
    u2 = vec @ W(O0 + "vec2_proj").T
    pre1 = torch.cat([xf, n1], -1) @ W(O0 + "update_net.layers.0").T + B(O0 + "update_net.layers.0")
    y = silu(pre1) @ W(O0 + "update_net.layers.2").T + B(O0 + "update_net.layers.2")
    F2 = F // 2
    xs, gate = silu(y[:, :F2]), y[:, F2:]
    vq = gate[:, None, :] * u2
    w1 = vq @ W(O1 + "vec1_proj").T
    n2 = torch.sqrt((w1 * w1).sum(1))
    pre2 = torch.cat([xs, n2], -1) @ W(O1 + "update_net.layers.0").T + B(O1 + "update_net.layers.0")
    y2 = silu(pre2) @ W(O1 + "update_net.layers.2").T + B(O1 + "update_net.layers.2")
    e = y2[:, :1] * sd.get("std", torch.ones((), dtype=dt))
    if atomref is not None:
        e = e + atomref[z]
    nmol = int(batch.max()) + 1
    Emol = torch.zeros(nmol, 1, dtype=dt).index_add(0, batch, e) + sd.get("mean", torch.zeros((), dtype=dt))

    # ---------------- reverse
    def safe_dir(u, n):  # u / n with 0 where n == 0 (reference masks zero rows out of the norm)
        return torch.where(n[:, None, :] > 0, u / torch.where(n > 0, n, torch.ones_like(n))[:, None, :], torch.zeros_like(u))

    g_y2 = torch.zeros_like(y2)
    g_y2[:, 0] = sd.get("std", torch.ones((), dtype=dt))
    g_pre2 = (g_y2 @ W(O1 + "update_net.layers.2")) * dsilu(pre2)
    g_h2 = g_pre2 @ W(O1 + "update_net.layers.0")
    g_xs, g_n2 = g_h2[:, :F2], g_h2[:, F2:]
    g_w1 = g_n2[:, None, :] * safe_dir(w1, n2)
    g_vq = g_w1 @ W(O1 + "vec1_proj")
    g_gate = (g_vq * u2).sum(1)
    g_u2 = g_vq * gate[:, None, :]
    g_y = torch.cat([g_xs * dsilu(y[:, :F2]), g_gate], -1)
    g_pre1 = (g_y @ W(O0 + "update_net.layers.2")) * dsilu(pre1)
    g_h1 = g_pre1 @ W(O0 + "update_net.layers.0")
    g_xf, g_n1 = g_h1[:, :F], g_h1[:, F:]
    g_u1 = g_n1[:, None, :] * safe_dir(u1, n1)
    g_vec = g_u1 @ W(O0 + "vec1_proj") + g_u2 @ W(O0 + "vec2_proj")
    g_x = layer_norm_bwd(g_xf, xfh, rstdf, W(R + "out_norm"))
    cache["g_x_final"], cache["g_vec_final"] = g_x, g_vec

    g_d = torch.zeros(E, dtype=dt)  # per DIRECTED edge here; the kernels add the two directions of a pair
    g_rhat = torch.zeros(E, 3, dtype=dt)
    for l in reversed(range(L)):
        Lp = R + f"attention_layers.{l}."
        S = saved[l]
        g_dx, g_dvec = g_x, g_vec  # residual: g_x / g_vec also flow straight through
        g_vdot = g_dx * S["o2"]
        g_o = torch.cat([(g_dvec * S["vec3"]).sum(1), g_dx * S["vdot"], g_dx], -1)
        g_vec3 = g_dvec * S["o1"][:, None, :]
        g_vagg = g_dvec
        g_xagg = g_o @ W(Lp + "o_proj")
        # ---- edge sweep
        a, cv, ca = S["a"], S["cv"], S["ca"]
        A = silu(a) * ca[:, None]
        Ac = A[:, hix]
        sx = S["vx"][src] * cv[:, None] * S["dvx"]
        s1 = S["v1"][src] * cv[:, None] * S["dv1"]
        s2 = S["v2"][src] * cv[:, None] * S["dv2"]
        gxa_t, gva_t = g_xagg[tgt], g_vagg[tgt]  # [E,F], [E,3,F]
        g_sx = gxa_t * Ac
        g_A = torch.zeros(E, H, dtype=dt).index_add(1, hix, gxa_t * sx)
        g_s1 = (gva_t * S["vec"][src]).sum(1)
        g_s2 = (gva_t * rhat[:, :, None]).sum(1)
        g_rhat += (gva_t * s2[:, None, :]).sum(2)
        g_vec_src = gva_t * s1[:, None, :]  # -> source
        # values: s = v[src] * cv * dv
        g_vx_src, g_v1_src, g_v2_src = g_sx * cv[:, None] * S["dvx"], g_s1 * cv[:, None] * S["dv1"], g_s2 * cv[:, None] * S["dv2"]
        vxs, v1s, v2s = S["vx"][src], S["v1"][src], S["v2"][src]
        g_d += ((g_sx * vxs * S["ddvx"] + g_s1 * v1s * S["ddv1"] + g_s2 * v2s * S["ddv2"]).sum(1)) * cv
        g_cv = (g_sx * vxs * S["dvx"] + g_s1 * v1s * S["dv1"] + g_s2 * v2s * S["dv2"]).sum(1)
        # attention
        g_a = g_A * dsilu(a) * ca[:, None]
        g_ca = (g_A * silu(a)).sum(1)
        g_ac = g_a[:, hix]
        qt, ks = S["q"][tgt], S["k"][src]
        g_q = torch.zeros(N, F, dtype=dt).index_add(0, tgt, g_ac * ks * S["dk"])
        g_k = torch.zeros(N, F, dtype=dt).index_add(0, src, g_ac * qt * S["dk"])
        g_d += (g_ac * qt * ks * S["ddk"]).sum(1)
        g_d += (g_cv if vc else g_ca) * dC
        g_v = torch.zeros(N, 3 * F, dtype=dt)
        g_v[:, S["ix"]] = torch.zeros(N, F, dtype=dt).index_add(0, src, g_vx_src)
        g_v[:, S["ix"] + hd] = torch.zeros(N, F, dtype=dt).index_add(0, src, g_v1_src)
        g_v[:, S["ix"] + 2 * hd] = torch.zeros(N, F, dtype=dt).index_add(0, src, g_v2_src)
        g_vec_in = torch.zeros(N, 3, F, dtype=dt).index_add(0, src, g_vec_src)
        # ---- node side
        g_vp = torch.cat([g_vdot[:, None, :] * S["vec2"], g_vdot[:, None, :] * S["vec1"], g_vec3], -1)
        g_vec = g_vec + g_vec_in + g_vp @ W(Lp + "vec_proj")
        g_xt = g_q @ W(Lp + "q_proj") + g_k @ W(Lp + "k_proj") + g_v @ W(Lp + "v_proj")
        g_x = g_x + layer_norm_bwd(g_xt, S["xh"], S["rstd"], W(Lp + "layernorm"))
        cache[f"g_x_layer{l}"], cache[f"g_vec_layer{l}"] = g_x, g_vec
    if ne:
        g_xn = g_x @ Wc[:, F:]
        g_Wn = torch.where(nself[:, None], g_xn[src] * embN[tgt], torch.zeros_like(Wn))
        g_d += (g_Wn * dWn).sum(1)
    # geometry: rhat = delta/d, d = |delta|, delta = pos[src] - pos[tgt]; self edges contribute nothing
    dsafe = torch.where(nself, d, torch.ones_like(d))
    g_delta = g_d[:, None] * rhat + (g_rhat - (g_rhat * rhat).sum(1, keepdim=True) * rhat) / dsafe[:, None]
    g_delta = torch.where(nself[:, None], g_delta, torch.zeros_like(g_delta))
    g_pos = torch.zeros(N, 3, dtype=dt).index_add(0, src, g_delta).index_add(0, tgt, -g_delta)
    if want_cache:
        return Emol, -g_pos, cache
    return Emol, -g_pos
