"""TEST INFRASTRUCTURE ONLY -- CPU specification of the SECOND-ORDER pass of force-matching training on the Equivariant Transformer:

    s(theta) = v . d(sum_m E_m)/d pos ,  wanted: d s / d theta  (and H v = d s / d pos)

written by hand, without autograd, as the forward-mode tangent (direction v in the positions) of the hand-derived E + F program
of oracle/et_adjoint.py - the same construction as oracle/tensornet_second_order.py for TensorNet.  The reference gets these
numbers from a second autograd pass (torchmdnet/models/model.py:618-628, create_graph=self.training) over torchmd_et.py:188-426 and
models/utils.py:83-117, 583-655.  tests/test_oracle.py pins this file to autograd-of-autograd over oracle/et_torch.py in fp64.
It is the specification for an engine pass that is NOT built yet (the engine differentiates the Equivariant Transformer's force
term by a difference quotient): gradient entries are returned by state-dict key.

Naming: a_t = tangent of a forward quantity, g_a = adjoint (seed 1 on every atom's energy), g_a_t = its tangent; a weight gradient
g_W = g_y^T x becomes g_y_t^T x + g_y^T x_t.
"""
import math

import torch

from oracle.tensornet_torch import neighbor_pairs


def silu(x):
    return x * torch.sigmoid(x)


def d1(x):  # silu'
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def d2(x):  # silu''
    s = torch.sigmoid(x)
    return s * (1 - s) * (2 + x * (1 - 2 * s))


def radial2(d, means, betas, lo, up):
    """phi[K], phi', phi'', C, C', C'' (models/utils.py:402-407, 506-528; lower cutoff 0 as in the configurations of the path)."""
    assert lo == 0.0
    alpha = 5.0 / (up - lo)
    kc = math.pi / up
    m = (d < up).to(d.dtype)
    c0, c1, c2 = 0.5 * (torch.cos(d * kc) + 1.0) * m, -0.5 * torch.sin(d * kc) * kc * m, -0.5 * torch.cos(d * kc) * kc * kc * m
    u = torch.exp(-alpha * (d - lo))[:, None]
    u1, u2 = -alpha * u, alpha * alpha * u
    g = torch.exp(-betas * (u - means) ** 2)
    h = -2 * betas * (u - means) * u1
    h1 = -2 * betas * (u1 * u1 + (u - means) * u2)
    g1, g2 = g * h, g * (h * h + h1)
    c0_, c1_, c2_ = c0[:, None], c1[:, None], c2[:, None]
    return c0_ * g, c1_ * g + c0_ * g1, c2_ * g + 2 * c1_ * g1 + c0_ * g2, c0, c1, c2


def ln_dual(x, x_t, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    xh = (x - mu) * rstd
    m = (xh * x_t).mean(-1, keepdim=True)
    xh_t = rstd * (x_t - x_t.mean(-1, keepdim=True) - xh * m)
    return xh * w + b, xh, rstd, xh_t * w, xh_t, -rstd * rstd * m


def ln_bwd_dual(g, g_t, xh, xh_t, rstd, rstd_t, w):
    gw, gw_t = g * w, g_t * w
    c1, c2 = gw.mean(-1, keepdim=True), (gw * xh).mean(-1, keepdim=True)
    c1_t, c2_t = gw_t.mean(-1, keepdim=True), (gw_t * xh + gw * xh_t).mean(-1, keepdim=True)
    core = gw - c1 - xh * c2
    return core * rstd, (gw_t - c1_t - xh_t * c2 - xh * c2_t) * rstd + core * rstd_t


def norm_dual(u, u_t):
    """n = |u| over the spatial axis (dim 1), 0 rows masked as the reference does (models/utils.py:640-646)."""
    n = torch.sqrt((u * u).sum(1))
    ok = n > 0
    ns = torch.where(ok, n, torch.ones_like(n))
    return n, torch.where(ok, (u * u_t).sum(1) / ns, torch.zeros_like(n))


def dir_dual(u, u_t, n, n_t):
    """u / n and its tangent, 0 where n == 0."""
    ok = (n > 0)[:, None, :]
    ns = torch.where(n > 0, n, torch.ones_like(n))[:, None, :]
    dirn = torch.where(ok, u / ns, torch.zeros_like(u))
    return dirn, torch.where(ok, (u_t - dirn * n_t[:, None, :]) / ns, torch.zeros_like(u))


def force_term(sd, hp, z, pos, batch, v, box=None, ge=None):
    """-> dict(E, F, s, grads={state-dict key: d s / d parameter}, Hv)."""
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
    ns1 = nself[:, None]
    dsafe = torch.where(nself, d, torch.ones_like(d))
    rhat = torch.where(ns1, delta / dsafe[:, None], torch.zeros_like(delta))
    delta_t = torch.where(ns1, v[src] - v[tgt], torch.zeros_like(delta))
    d_t = (rhat * delta_t).sum(1)
    rhat_t = torch.where(ns1, (delta_t - rhat * d_t[:, None]) / dsafe[:, None], torch.zeros_like(delta))
    phi, dphi, d2phi, C, dC, d2C = radial2(d, sd[R + "distance_expansion.means"], sd[R + "distance_expansion.betas"], lo, up)
    phi_t, C_t = dphi * d_t[:, None], dC * d_t
    G = {}  # d s / d parameter

    def W(k):
        return sd[k + ".weight"]

    def B(k):
        return sd[k + ".bias"]

    def lin_grad(key, gy, gy_t, x, x_t, bias=True):
        """y = x W^T (+ b): d s / d W = g_y_t^T x + g_y^T x_t, d s / d b = colsum(g_y_t); rows may carry a spatial axis."""
        gy2, gyt2, x2, xt2 = (t.reshape(-1, t.shape[-1]) for t in (gy, gy_t, x, x_t))
        G[key + ".weight"] = G.get(key + ".weight", 0) + gyt2.t() @ x2 + gy2.t() @ xt2
        if bias:
            G[key + ".bias"] = G.get(key + ".bias", 0) + gyt2.sum(0)

    def scat(idx, vals, shape):
        return torch.zeros(shape, dtype=dt).index_add(0, idx, vals)

    # ================= forward with tangents
    Wemb = W(R + "embedding")
    x0 = Wemb[z]
    ne = hp["neighbor_embedding"]
    if ne:
        Np = R + "neighbor_embedding."
        Wdp = W(Np + "distance_proj")
        e_n, e_n_t = phi @ Wdp.T + B(Np + "distance_proj"), phi_t @ Wdp.T
        Wn, Wn_t = e_n * C[:, None], e_n_t * C[:, None] + e_n * C_t[:, None]
        embN = W(Np + "embedding")[z]
        msg = torch.where(ns1, Wn * embN[tgt], torch.zeros_like(Wn))
        msg_t = torch.where(ns1, Wn_t * embN[tgt], torch.zeros_like(Wn))
        xn, xn_t = scat(src, msg, (N, F)), scat(src, msg_t, (N, F))
        Wc = W(Np + "combine")
        x = x0 @ Wc[:, :F].T + xn @ Wc[:, F:].T + B(Np + "combine")
        x_t = xn_t @ Wc[:, F:].T
    else:
        x, x_t = x0, torch.zeros_like(x0)
    vec, vec_t = torch.zeros(N, 3, F, dtype=dt), torch.zeros(N, 3, F, dtype=dt)
    hix = torch.arange(F) // hd
    cidx = torch.arange(F)
    ix = (cidx // hd) * 3 * hd + cidx % hd  # channel c of the three value thirds lives at v[:, ix + {0, hd, 2 hd}]
    cv, cv_t = (C, C_t) if vc else (torch.ones_like(C), torch.zeros_like(C))
    ca, ca_t = (torch.ones_like(C), torch.zeros_like(C)) if vc else (C, C_t)
    dcv, d2cv = (dC, d2C) if vc else (torch.zeros_like(C), torch.zeros_like(C))
    dca, d2ca = (torch.zeros_like(C), torch.zeros_like(C)) if vc else (dC, d2C)
    saved = []
    for l in range(L):
        Lp = R + f"attention_layers.{l}."
        S = {}
        S["x"], S["x_t"], S["vec"], S["vec_t"] = x, x_t, vec, vec_t
        xt, xh, rstd, xt_t, xh_t, rstd_t = ln_dual(x, x_t, W(Lp + "layernorm"), B(Lp + "layernorm"))
        q, q_t = xt @ W(Lp + "q_proj").T + B(Lp + "q_proj"), xt_t @ W(Lp + "q_proj").T
        k, k_t = xt @ W(Lp + "k_proj").T + B(Lp + "k_proj"), xt_t @ W(Lp + "k_proj").T
        vv, vv_t = xt @ W(Lp + "v_proj").T + B(Lp + "v_proj"), xt_t @ W(Lp + "v_proj").T
        vp, vp_t = vec @ W(Lp + "vec_proj").T, vec_t @ W(Lp + "vec_proj").T
        vec1, vec2, vec3 = vp[..., :F], vp[..., F:2 * F], vp[..., 2 * F:]
        vec1_t, vec2_t, vec3_t = vp_t[..., :F], vp_t[..., F:2 * F], vp_t[..., 2 * F:]
        vdot, vdot_t = (vec1 * vec2).sum(1), (vec1_t * vec2 + vec1 * vec2_t).sum(1)
        has_dk, has_dv = (Lp + "dk_proj.weight") in sd, (Lp + "dv_proj.weight") in sd
        if has_dk:
            ek, ek_t = phi @ W(Lp + "dk_proj").T + B(Lp + "dk_proj"), phi_t @ W(Lp + "dk_proj").T
            dk, dk_t = silu(ek), d1(ek) * ek_t
        else:
            ek = ek_t = None
            dk, dk_t = torch.ones(E, F, dtype=dt), torch.zeros(E, F, dtype=dt)
        if has_dv:
            ev, ev_t = phi @ W(Lp + "dv_proj").T + B(Lp + "dv_proj"), phi_t @ W(Lp + "dv_proj").T
            dv, dv_t = silu(ev), d1(ev) * ev_t
        else:
            ev = ev_t = None
            dv, dv_t = torch.ones(E, 3 * F, dtype=dt), torch.zeros(E, 3 * F, dtype=dt)
        # the three thirds (x, 1, 2) of values and value filters, per channel
        v3 = [vv[:, ix + j * hd] for j in range(3)]
        v3_t = [vv_t[:, ix + j * hd] for j in range(3)]
        f3 = [dv[:, ix + j * hd] for j in range(3)]
        f3_t = [dv_t[:, ix + j * hd] for j in range(3)]
        qk = q[tgt] * k[src]
        qk_t = q_t[tgt] * k[src] + q[tgt] * k_t[src]
        a = torch.zeros(E, H, dtype=dt).index_add(1, hix, qk * dk)
        a_t = torch.zeros(E, H, dtype=dt).index_add(1, hix, qk_t * dk + qk * dk_t)
        A, A_t = silu(a) * ca[:, None], d1(a) * a_t * ca[:, None] + silu(a) * ca_t[:, None]
        Ac, Ac_t = A[:, hix], A_t[:, hix]
        s3 = [v3[j][src] * cv[:, None] * f3[j] for j in range(3)]
        s3_t = [v3_t[j][src] * cv[:, None] * f3[j] + v3[j][src] * cv_t[:, None] * f3[j] + v3[j][src] * cv[:, None] * f3_t[j] for j in range(3)]
        xagg, xagg_t = scat(tgt, s3[0] * Ac, (N, F)), scat(tgt, s3_t[0] * Ac + s3[0] * Ac_t, (N, F))
        vm = vec[src] * s3[1][:, None, :] + s3[2][:, None, :] * rhat[:, :, None]
        vm_t = (vec_t[src] * s3[1][:, None, :] + vec[src] * s3_t[1][:, None, :] + s3_t[2][:, None, :] * rhat[:, :, None]
                + s3[2][:, None, :] * rhat_t[:, :, None])
        vagg, vagg_t = scat(tgt, vm, (N, 3, F)), scat(tgt, vm_t, (N, 3, F))
        o, o_t = xagg @ W(Lp + "o_proj").T + B(Lp + "o_proj"), xagg_t @ W(Lp + "o_proj").T
        o1, o2, o3 = o[:, :F], o[:, F:2 * F], o[:, 2 * F:]
        o1_t, o2_t, o3_t = o_t[:, :F], o_t[:, F:2 * F], o_t[:, 2 * F:]
        dx, dx_t = vdot * o2 + o3, vdot_t * o2 + vdot * o2_t + o3_t
        dvec = vec3 * o1[:, None, :] + vagg
        dvec_t = vec3_t * o1[:, None, :] + vec3 * o1_t[:, None, :] + vagg_t
        S.update(xt=xt, xt_t=xt_t, xh=xh, xh_t=xh_t, rstd=rstd, rstd_t=rstd_t, q=q, q_t=q_t, k=k, k_t=k_t, v3=v3, v3_t=v3_t, f3=f3, f3_t=f3_t,
                 vec1=vec1, vec1_t=vec1_t, vec2=vec2, vec2_t=vec2_t, vec3=vec3, vec3_t=vec3_t, vdot=vdot, vdot_t=vdot_t, dk=dk, dk_t=dk_t,
                 ek=ek, ek_t=ek_t, ev=ev, ev_t=ev_t, a=a, a_t=a_t, s3=s3, s3_t=s3_t, Ac=Ac, Ac_t=Ac_t, xagg=xagg, xagg_t=xagg_t,
                 o1=o1, o1_t=o1_t, o2=o2, o2_t=o2_t, has_dk=has_dk, has_dv=has_dv)
        saved.append(S)
        x, x_t, vec, vec_t = x + dx, x_t + dx_t, vec + dvec, vec_t + dvec_t
    xf, xfh, rstdf, xf_t, xfh_t, rstdf_t = ln_dual(x, x_t, W(R + "out_norm"), B(R + "out_norm"))

    # head (models/utils.py:626-655, output_modules.py:120-163)
    O0, O1 = "output_model.output_network.0.", "output_model.output_network.1."
    F2 = F // 2
    std = sd.get("std", torch.ones((), dtype=dt))
    u1, u1_t = vec @ W(O0 + "vec1_proj").T, vec_t @ W(O0 + "vec1_proj").T
    n1, n1_t = norm_dual(u1, u1_t)
    u2, u2_t = vec @ W(O0 + "vec2_proj").T, vec_t @ W(O0 + "vec2_proj").T
    in1, in1_t = torch.cat([xf, n1], -1), torch.cat([xf_t, n1_t], -1)
    pre1, pre1_t = in1 @ W(O0 + "update_net.layers.0").T + B(O0 + "update_net.layers.0"), in1_t @ W(O0 + "update_net.layers.0").T
    h1, h1_t = silu(pre1), d1(pre1) * pre1_t
    y, y_t = h1 @ W(O0 + "update_net.layers.2").T + B(O0 + "update_net.layers.2"), h1_t @ W(O0 + "update_net.layers.2").T
    ys, ys_t = y[:, :F2], y_t[:, :F2]
    xs, xs_t = silu(ys), d1(ys) * ys_t
    gate, gate_t = y[:, F2:], y_t[:, F2:]
    vq, vq_t = gate[:, None, :] * u2, gate_t[:, None, :] * u2 + gate[:, None, :] * u2_t
    w1, w1_t = vq @ W(O1 + "vec1_proj").T, vq_t @ W(O1 + "vec1_proj").T
    n2, n2_t = norm_dual(w1, w1_t)
    in2, in2_t = torch.cat([xs, n2], -1), torch.cat([xs_t, n2_t], -1)
    pre2, pre2_t = in2 @ W(O1 + "update_net.layers.0").T + B(O1 + "update_net.layers.0"), in2_t @ W(O1 + "update_net.layers.0").T
    h2, h2_t = silu(pre2), d1(pre2) * pre2_t
    y2, y2_t = h2 @ W(O1 + "update_net.layers.2").T + B(O1 + "update_net.layers.2"), h2_t @ W(O1 + "update_net.layers.2").T
    e, e_t = y2[:, :1] * std, y2_t[:, :1] * std
    nmol = int(batch.max()) + 1
    Emol = torch.zeros(nmol, 1, dtype=dt).index_add(0, batch, e) + sd.get("mean", torch.zeros((), dtype=dt))
    s_val = e_t.sum()

    # ================= reverse with tangents: head
    g_y2, g_y2_t = torch.zeros_like(y2), torch.zeros_like(y2)
    g_y2[:, 0] = std
    if ge is not None:  # one-pass training: gradient of S = s - sum_m ge_m E_m (oracle/tensornet_second_order.py force_term)
        g_y2_t[:, 0] = -std * ge.to(dt).reshape(-1)[batch]
        s_val = s_val - (ge.to(dt).reshape(-1) * Emol.reshape(-1)).sum()
    lin_grad(O1 + "update_net.layers.2", g_y2, g_y2_t, h2, h2_t)
    g_h2, g_h2_t = g_y2 @ W(O1 + "update_net.layers.2"), g_y2_t @ W(O1 + "update_net.layers.2")
    g_pre2, g_pre2_t = g_h2 * d1(pre2), g_h2_t * d1(pre2) + g_h2 * d2(pre2) * pre2_t
    lin_grad(O1 + "update_net.layers.0", g_pre2, g_pre2_t, in2, in2_t)
    g_in2, g_in2_t = g_pre2 @ W(O1 + "update_net.layers.0"), g_pre2_t @ W(O1 + "update_net.layers.0")
    g_xs, g_n2, g_xs_t, g_n2_t = g_in2[:, :F2], g_in2[:, F2:], g_in2_t[:, :F2], g_in2_t[:, F2:]
    dir2, dir2_t = dir_dual(w1, w1_t, n2, n2_t)
    g_w1, g_w1_t = g_n2[:, None, :] * dir2, g_n2_t[:, None, :] * dir2 + g_n2[:, None, :] * dir2_t
    lin_grad(O1 + "vec1_proj", g_w1, g_w1_t, vq, vq_t, bias=False)
    g_vq, g_vq_t = g_w1 @ W(O1 + "vec1_proj"), g_w1_t @ W(O1 + "vec1_proj")
    g_gate, g_gate_t = (g_vq * u2).sum(1), (g_vq_t * u2 + g_vq * u2_t).sum(1)
    g_u2, g_u2_t = g_vq * gate[:, None, :], g_vq_t * gate[:, None, :] + g_vq * gate_t[:, None, :]
    g_ys, g_ys_t = g_xs * d1(ys), g_xs_t * d1(ys) + g_xs * d2(ys) * ys_t
    g_y, g_y_t = torch.cat([g_ys, g_gate], -1), torch.cat([g_ys_t, g_gate_t], -1)
    lin_grad(O0 + "update_net.layers.2", g_y, g_y_t, h1, h1_t)
    g_h1, g_h1_t = g_y @ W(O0 + "update_net.layers.2"), g_y_t @ W(O0 + "update_net.layers.2")
    g_pre1, g_pre1_t = g_h1 * d1(pre1), g_h1_t * d1(pre1) + g_h1 * d2(pre1) * pre1_t
    lin_grad(O0 + "update_net.layers.0", g_pre1, g_pre1_t, in1, in1_t)
    g_in1, g_in1_t = g_pre1 @ W(O0 + "update_net.layers.0"), g_pre1_t @ W(O0 + "update_net.layers.0")
    g_xf, g_n1, g_xf_t, g_n1_t = g_in1[:, :F], g_in1[:, F:], g_in1_t[:, :F], g_in1_t[:, F:]
    dir1, dir1_t = dir_dual(u1, u1_t, n1, n1_t)
    g_u1, g_u1_t = g_n1[:, None, :] * dir1, g_n1_t[:, None, :] * dir1 + g_n1[:, None, :] * dir1_t
    lin_grad(O0 + "vec1_proj", g_u1, g_u1_t, vec, vec_t, bias=False)
    lin_grad(O0 + "vec2_proj", g_u2, g_u2_t, vec, vec_t, bias=False)
    g_vec = g_u1 @ W(O0 + "vec1_proj") + g_u2 @ W(O0 + "vec2_proj")
    g_vec_t = g_u1_t @ W(O0 + "vec1_proj") + g_u2_t @ W(O0 + "vec2_proj")
    G[R + "out_norm.weight"] = (g_xf_t * xfh + g_xf * xfh_t).sum(0)
    G[R + "out_norm.bias"] = g_xf_t.sum(0)
    g_x, g_x_t = ln_bwd_dual(g_xf, g_xf_t, xfh, xfh_t, rstdf, rstdf_t, W(R + "out_norm"))

    # ================= reverse with tangents: attention layers
    g_d, g_d_t = torch.zeros(E, dtype=dt), torch.zeros(E, dtype=dt)  # per DIRECTED edge
    g_rhat, g_rhat_t = torch.zeros(E, 3, dtype=dt), torch.zeros(E, 3, dtype=dt)
    for l in reversed(range(L)):
        Lp = R + f"attention_layers.{l}."
        S = saved[l]
        vec_l, vec_l_t = S["vec"], S["vec_t"]
        g_dx, g_dx_t, g_dvec, g_dvec_t = g_x, g_x_t, g_vec, g_vec_t
        g_vdot, g_vdot_t = g_dx * S["o2"], g_dx_t * S["o2"] + g_dx * S["o2_t"]
        g_o = torch.cat([(g_dvec * S["vec3"]).sum(1), g_dx * S["vdot"], g_dx], -1)
        g_o_t = torch.cat([(g_dvec_t * S["vec3"] + g_dvec * S["vec3_t"]).sum(1), g_dx_t * S["vdot"] + g_dx * S["vdot_t"], g_dx_t], -1)
        g_vec3, g_vec3_t = g_dvec * S["o1"][:, None, :], g_dvec_t * S["o1"][:, None, :] + g_dvec * S["o1_t"][:, None, :]
        lin_grad(Lp + "o_proj", g_o, g_o_t, S["xagg"], S["xagg_t"])
        g_xagg, g_xagg_t = g_o @ W(Lp + "o_proj"), g_o_t @ W(Lp + "o_proj")
        # ---- edge sweep
        a, a_t, Ac, Ac_t, s3, s3_t = S["a"], S["a_t"], S["Ac"], S["Ac_t"], S["s3"], S["s3_t"]
        v3, v3_t, f3, f3_t = S["v3"], S["v3_t"], S["f3"], S["f3_t"]
        gxa, gxa_t, gva, gva_t = g_xagg[tgt], g_xagg_t[tgt], g_dvec[tgt], g_dvec_t[tgt]
        g_s = [gxa * Ac, (gva * vec_l[src]).sum(1), (gva * rhat[:, :, None]).sum(1)]
        g_s_t = [gxa_t * Ac + gxa * Ac_t, (gva_t * vec_l[src] + gva * vec_l_t[src]).sum(1),
                 (gva_t * rhat[:, :, None] + gva * rhat_t[:, :, None]).sum(1)]
        g_A = torch.zeros(E, H, dtype=dt).index_add(1, hix, gxa * s3[0])
        g_A_t = torch.zeros(E, H, dtype=dt).index_add(1, hix, gxa_t * s3[0] + gxa * s3_t[0])
        g_rhat += (gva * s3[2][:, None, :]).sum(2)
        g_rhat_t += (gva_t * s3[2][:, None, :] + gva * s3_t[2][:, None, :]).sum(2)
        g_vec_src, g_vec_src_t = gva * s3[1][:, None, :], gva_t * s3[1][:, None, :] + gva * s3_t[1][:, None, :]
        # values and value filters: s_j = v_j[src] cv f_j
        g_v = torch.zeros(N, 3 * F, dtype=dt)
        g_v_t = torch.zeros(N, 3 * F, dtype=dt)
        g_dv, g_dv_t = torch.zeros(E, 3 * F, dtype=dt), torch.zeros(E, 3 * F, dtype=dt)  # adjoint of the filter values dv
        g_cv, g_cv_t = torch.zeros(E, dtype=dt), torch.zeros(E, dtype=dt)
        for j in range(3):
            vs, vs_t = v3[j][src], v3_t[j][src]
            gv_src = g_s[j] * cv[:, None] * f3[j]
            gv_src_t = g_s_t[j] * cv[:, None] * f3[j] + g_s[j] * cv_t[:, None] * f3[j] + g_s[j] * cv[:, None] * f3_t[j]
            g_v[:, ix + j * hd] = scat(src, gv_src, (N, F))
            g_v_t[:, ix + j * hd] = scat(src, gv_src_t, (N, F))
            g_dv[:, ix + j * hd] = g_s[j] * vs * cv[:, None]
            g_dv_t[:, ix + j * hd] = g_s_t[j] * vs * cv[:, None] + g_s[j] * vs_t * cv[:, None] + g_s[j] * vs * cv_t[:, None]
            g_cv += (g_s[j] * vs * f3[j]).sum(1)
            g_cv_t += (g_s_t[j] * vs * f3[j] + g_s[j] * vs_t * f3[j] + g_s[j] * vs * f3_t[j]).sum(1)
        # attention weights
        g_a, g_a_t = g_A * d1(a) * ca[:, None], g_A_t * d1(a) * ca[:, None] + g_A * d2(a) * a_t * ca[:, None] + g_A * d1(a) * ca_t[:, None]
        g_ca, g_ca_t = (g_A * silu(a)).sum(1), (g_A_t * silu(a) + g_A * d1(a) * a_t).sum(1)
        g_ac, g_ac_t = g_a[:, hix], g_a_t[:, hix]
        qt_, qt_t, ks, ks_t, dk, dk_t = S["q"][tgt], S["q_t"][tgt], S["k"][src], S["k_t"][src], S["dk"], S["dk_t"]
        g_q = scat(tgt, g_ac * ks * dk, (N, F))
        g_q_t = scat(tgt, g_ac_t * ks * dk + g_ac * ks_t * dk + g_ac * ks * dk_t, (N, F))
        g_k = scat(src, g_ac * qt_ * dk, (N, F))
        g_k_t = scat(src, g_ac_t * qt_ * dk + g_ac * qt_t * dk + g_ac * qt_ * dk_t, (N, F))
        g_dk, g_dk_t = g_ac * qt_ * ks, g_ac_t * qt_ * ks + g_ac * qt_t * ks + g_ac * qt_ * ks_t  # adjoint of the key filter dk
        # distance filters: dk = silu(ek), ek = phi Wdk^T + b  ->  weights, and the distance gradient through phi
        g_phi, g_phi_t = torch.zeros(E, phi.shape[1], dtype=dt), torch.zeros(E, phi.shape[1], dtype=dt)
        if S["has_dk"]:
            ek, ek_t = S["ek"], S["ek_t"]
            g_ek, g_ek_t = g_dk * d1(ek), g_dk_t * d1(ek) + g_dk * d2(ek) * ek_t
            lin_grad(Lp + "dk_proj", g_ek, g_ek_t, phi, phi_t)
            g_phi, g_phi_t = g_phi + g_ek @ W(Lp + "dk_proj"), g_phi_t + g_ek_t @ W(Lp + "dk_proj")
        if S["has_dv"]:
            ev, ev_t = S["ev"], S["ev_t"]
            g_ev, g_ev_t = g_dv * d1(ev), g_dv_t * d1(ev) + g_dv * d2(ev) * ev_t
            lin_grad(Lp + "dv_proj", g_ev, g_ev_t, phi, phi_t)
            g_phi, g_phi_t = g_phi + g_ev @ W(Lp + "dv_proj"), g_phi_t + g_ev_t @ W(Lp + "dv_proj")
        g_d += (g_phi * dphi).sum(1) + g_cv * dcv + g_ca * dca
        g_d_t += ((g_phi_t * dphi).sum(1) + (g_phi * d2phi).sum(1) * d_t + g_cv_t * dcv + g_cv * d2cv * d_t + g_ca_t * dca
                  + g_ca * d2ca * d_t)
        g_vec_in, g_vec_in_t = scat(src, g_vec_src, (N, 3, F)), scat(src, g_vec_src_t, (N, 3, F))
        # ---- node side
        g_vp = torch.cat([g_vdot[:, None, :] * S["vec2"], g_vdot[:, None, :] * S["vec1"], g_vec3], -1)
        g_vp_t = torch.cat([g_vdot_t[:, None, :] * S["vec2"] + g_vdot[:, None, :] * S["vec2_t"],
                            g_vdot_t[:, None, :] * S["vec1"] + g_vdot[:, None, :] * S["vec1_t"], g_vec3_t], -1)
        lin_grad(Lp + "vec_proj", g_vp, g_vp_t, vec_l, vec_l_t, bias=False)
        g_vec = g_vec + g_vec_in + g_vp @ W(Lp + "vec_proj")
        g_vec_t = g_vec_t + g_vec_in_t + g_vp_t @ W(Lp + "vec_proj")
        lin_grad(Lp + "q_proj", g_q, g_q_t, S["xt"], S["xt_t"])
        lin_grad(Lp + "k_proj", g_k, g_k_t, S["xt"], S["xt_t"])
        lin_grad(Lp + "v_proj", g_v, g_v_t, S["xt"], S["xt_t"])
        g_xt = g_q @ W(Lp + "q_proj") + g_k @ W(Lp + "k_proj") + g_v @ W(Lp + "v_proj")
        g_xt_t = g_q_t @ W(Lp + "q_proj") + g_k_t @ W(Lp + "k_proj") + g_v_t @ W(Lp + "v_proj")
        G[Lp + "layernorm.weight"] = (g_xt_t * S["xh"] + g_xt * S["xh_t"]).sum(0)
        G[Lp + "layernorm.bias"] = g_xt_t.sum(0)
        gl, gl_t = ln_bwd_dual(g_xt, g_xt_t, S["xh"], S["xh_t"], S["rstd"], S["rstd_t"], W(Lp + "layernorm"))
        g_x, g_x_t = g_x + gl, g_x_t + gl_t

    # ================= embeddings
    if ne:
        Np = R + "neighbor_embedding."
        cat_in, cat_in_t = torch.cat([x0, xn], -1), torch.cat([torch.zeros_like(x0), xn_t], -1)
        lin_grad(Np + "combine", g_x, g_x_t, cat_in, cat_in_t)
        g_x0, g_x0_t = g_x @ Wc[:, :F], g_x_t @ Wc[:, :F]
        g_xn, g_xn_t = g_x @ Wc[:, F:], g_x_t @ Wc[:, F:]
        g_msg = torch.where(ns1, g_xn[src], torch.zeros(E, F, dtype=dt))
        g_msg_t = torch.where(ns1, g_xn_t[src], torch.zeros(E, F, dtype=dt))
        g_Wn, g_Wn_t = g_msg * embN[tgt], g_msg_t * embN[tgt]
        g_embN, g_embN_t = scat(tgt, g_msg * Wn, (N, F)), scat(tgt, g_msg_t * Wn + g_msg * Wn_t, (N, F))
        G[Np + "embedding.weight"] = torch.zeros_like(W(Np + "embedding")).index_add(0, z, g_embN_t)
        g_en, g_en_t = g_Wn * C[:, None], g_Wn_t * C[:, None] + g_Wn * C_t[:, None]
        lin_grad(Np + "distance_proj", g_en, g_en_t, phi, phi_t)
        g_phin, g_phin_t = g_en @ Wdp, g_en_t @ Wdp
        g_Cn, g_Cn_t = (g_Wn * e_n).sum(1), (g_Wn_t * e_n + g_Wn * e_n_t).sum(1)
        g_d += (g_phin * dphi).sum(1) + g_Cn * dC
        g_d_t += (g_phin_t * dphi).sum(1) + (g_phin * d2phi).sum(1) * d_t + g_Cn_t * dC + g_Cn * d2C * d_t
    else:
        g_x0, g_x0_t = g_x, g_x_t
    G[R + "embedding.weight"] = torch.zeros_like(Wemb).index_add(0, z, g_x0_t)

    # ================= geometry: forces and H v   (rhat = delta / d, delta = pos[src] - pos[tgt]; self edges contribute nothing)
    a_ = (g_rhat * rhat).sum(1, keepdim=True)
    a_t = (g_rhat_t * rhat + g_rhat * rhat_t).sum(1, keepdim=True)
    t1 = (g_rhat - a_ * rhat) / dsafe[:, None]
    t1_t = (g_rhat_t - a_t * rhat - a_ * rhat_t) / dsafe[:, None] - t1 * (d_t / dsafe)[:, None]
    g_delta = torch.where(ns1, g_d[:, None] * rhat + t1, torch.zeros_like(delta))
    g_delta_t = torch.where(ns1, g_d_t[:, None] * rhat + g_d[:, None] * rhat_t + t1_t, torch.zeros_like(delta))
    g_pos = torch.zeros(N, 3, dtype=dt).index_add(0, src, g_delta).index_add(0, tgt, -g_delta)
    Hv = torch.zeros(N, 3, dtype=dt).index_add(0, src, g_delta_t).index_add(0, tgt, -g_delta_t)
    return dict(E=Emol, F=-g_pos, s=s_val, grads=G, Hv=Hv)
