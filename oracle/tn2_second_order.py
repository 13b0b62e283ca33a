"""TEST INFRASTRUCTURE ONLY -- CPU specification of the SECOND-ORDER pass of force-matching training on TensorNet2 +
ScalarPlusWeightedCoulomb:      s(theta) = v . d(sum_m E_m)/d pos ,  wanted: d s / d theta  and  H v = d s / d pos

written by hand, without autograd, as the forward-mode tangent (direction v in the positions) of a hand-derived E + F program of
the path (reference: torchmdnet/models/tensornet2.py:49-626, output_modules.py:323-609; its second autograd pass model.py:618-628).
It is oracle/tensornet_second_order.py (same layout, graph, naming; most statements are the same) plus what TensorNet2 adds:
the ChargePredict heads with their per-molecule equilibration after the embedding and after every layer, the edge MLP per DIRECTED
edge on [phi(d), c_i, c_j], no charge factor in the group product / update, and the damped pair Coulomb energy of all charge
channels in the head.  tests/test_oracle.py pins it to autograd-of-autograd over oracle/tn2_torch.py in fp64 (value half: E, F).
An engine pass built on it does not exist yet (TensorNet2 keeps the difference quotient).
"""
import math

import torch
import torch.nn.functional as Fn

from oracle import tensornet_adjoint as A
from oracle.tensornet_second_order import (TYPE_OF, cutoff_derivs, gA_of, ln_bwd_dual, ln_dual, r6_dual, r6_of, rbf_derivs, silu_d1, silu_d2,
                                            tlin, tlin_grad, tsum, u0_of)
from oracle.tensornet_torch import neighbor_pairs
from oracle.tn2_torch import COULOMB_FACTOR


def edge_gather(src, we, rows, cols):
    """out[i, c] = sum_{e: rows[e] = i} we[e, type(c)] src[cols[e], c]      (weights per DIRECTED edge)"""
    return torch.zeros_like(src).index_add(0, rows, we[:, TYPE_OF, :] * src[cols])


def edge_gather_T(g_out, we, rows, cols, n):
    """adjoint of edge_gather wrt src: g_src[j, c] = sum_{e: cols[e] = j} we[e, type(c)] g_out[rows[e], c]"""
    return torch.zeros_like(g_out).index_add(0, cols, we[:, TYPE_OF, :] * g_out[rows])


def _cp_layers(sd, pre):
    ks = [0]
    while (pre + f"q_mlp.layers.{ks[-1] + 2}.weight") in sd:
        ks.append(ks[-1] + 2)
    return ks


def charge_predict_dual(sd, pre, X, X_t, batch, Qatom, nmol, qd):
    """ChargePredict + equilibration (tensornet2.py:49-157) on the 9-component layout, with tangents.  Features: the trace component I
    itself (not 3 I^2), |A|^2 = 2 sum v^2, |S|^2."""
    n, _, Fh = X.shape
    dq, dq_t = A.dquad(X), A.dquad(X_t)
    xS = X[:, 4] ** 2 + X[:, 7] ** 2 + (X[:, 4] + X[:, 7]) ** 2 + 2 * (X[:, 5] ** 2 + X[:, 6] ** 2 + X[:, 8] ** 2)
    feat = torch.cat([X[:, 0], 2 * (X[:, 1:4] ** 2).sum(1), xS], -1)
    feat_t = torch.cat([X_t[:, 0], (dq[:, 1:4] * X_t[:, 1:4]).sum(1), (dq[:, 4:9] * X_t[:, 4:9]).sum(1)], -1)
    ln_w, ln_b = sd[pre + "q_norm.weight"], sd[pre + "q_norm.bias"]
    ln, xh, rstd, ln_t, xh_t, rstd_t = ln_dual(feat, feat_t, ln_w, ln_b)
    ks = _cp_layers(sd, pre)
    hs, hs_t, pres, pres_t = [ln], [ln_t], [], []
    for k in ks[:-1]:
        W, b = sd[pre + f"q_mlp.layers.{k}.weight"], sd[pre + f"q_mlp.layers.{k}.bias"]
        a, a_t = hs[-1] @ W.t() + b, hs_t[-1] @ W.t()
        pres.append(a)
        pres_t.append(a_t)
        hs.append(Fn.silu(a))
        hs_t.append(silu_d1(a) * a_t)
    W, b = sd[pre + f"q_mlp.layers.{ks[-1]}.weight"], sd[pre + f"q_mlp.layers.{ks[-1]}.bias"]
    out, out_t = hs[-1] @ W.t() + b, hs_t[-1] @ W.t()
    c, f, c_t, f_t = out[:, :qd], out[:, qd:], out_t[:, :qd], out_t[:, qd:]
    fu, fu_t = f ** 2, 2 * f * f_t
    msum = lambda t: torch.zeros(nmol, qd, dtype=X.dtype).index_add(0, batch, t)
    Fu, Fu_t = msum(fu) + 1.0e-6, msum(fu_t)
    Qu, Qu_t = msum(c), msum(c_t)
    dQ, dQ_t = Qatom.unsqueeze(-1) - Qu[batch], -Qu_t[batch]
    r, r_t = fu / Fu[batch], fu_t / Fu[batch] - fu * Fu_t[batch] / Fu[batch] ** 2
    ch, ch_t = c + r * dQ, c_t + r_t * dQ + r * dQ_t
    return dict(ch=ch, ch_t=ch_t, X=X, X_t=X_t, dq=dq, dq_t=dq_t, xh=xh, xh_t=xh_t, rstd=rstd, rstd_t=rstd_t, hs=hs, hs_t=hs_t, pres=pres,
                pres_t=pres_t, f=f, f_t=f_t, fu=fu, fu_t=fu_t, Fu=Fu, Fu_t=Fu_t, dQ=dQ, dQ_t=dQ_t, r=r, r_t=r_t, ks=ks)


def charge_predict_bwd_dual(sd, pre, S, g_ch, g_ch_t, batch, nmol, qd, G_):
    """adjoint of charge_predict_dual's head and its tangent: -> (g_X, g_X_t) [N, 9, F]; parameter entries into G_ by state-dict key"""
    n = g_ch.shape[0]
    dt = g_ch.dtype
    msum = lambda t: torch.zeros(nmol, qd, dtype=dt).index_add(0, batch, t)
    r, r_t, dQ, dQ_t, fu, fu_t, f, f_t = S["r"], S["r_t"], S["dQ"], S["dQ_t"], S["fu"], S["fu_t"], S["f"], S["f_t"]
    Fb, Fb_t = S["Fu"][batch], S["Fu_t"][batch]
    # ch = c + r dQ, dQ = Qatom - Qu[batch], Qu = sum_mol c, r = fu / Fu[batch], Fu = sum_mol fu + eps
    g_r, g_r_t = g_ch * dQ, g_ch_t * dQ + g_ch * dQ_t
    g_dQ, g_dQ_t = g_ch * r, g_ch_t * r + g_ch * r_t
    g_c, g_c_t = g_ch - msum(g_dQ)[batch], g_ch_t - msum(g_dQ_t)[batch]
    g_fu, g_fu_t = g_r / Fb, g_r_t / Fb - g_r * Fb_t / Fb ** 2
    hF, hF_t = g_r * fu / Fb ** 2, (g_r_t * fu + g_r * fu_t) / Fb ** 2 - 2 * g_r * fu * Fb_t / Fb ** 3
    g_fu, g_fu_t = g_fu - msum(hF)[batch], g_fu_t - msum(hF_t)[batch]
    g_f, g_f_t = 2 * f * g_fu, 2 * (f_t * g_fu + f * g_fu_t)
    g_out, g_out_t = torch.cat([g_c, g_f], -1), torch.cat([g_c_t, g_f_t], -1)
    ks, hs, hs_t, pres, pres_t = S["ks"], S["hs"], S["hs_t"], S["pres"], S["pres_t"]

    def lin_grad(key, gy, gy_t, x, x_t):
        G_[key + ".weight"] = G_.get(key + ".weight", 0) + gy_t.t() @ x + gy.t() @ x_t
        G_[key + ".bias"] = G_.get(key + ".bias", 0) + gy_t.sum(0)

    lin_grad(pre + f"q_mlp.layers.{ks[-1]}", g_out, g_out_t, hs[-1], hs_t[-1])
    W = sd[pre + f"q_mlp.layers.{ks[-1]}.weight"]
    g_h, g_h_t = g_out @ W, g_out_t @ W
    for i in reversed(range(len(ks) - 1)):
        a, a_t = pres[i], pres_t[i]
        g_a, g_a_t = g_h * silu_d1(a), g_h_t * silu_d1(a) + g_h * silu_d2(a) * a_t
        lin_grad(pre + f"q_mlp.layers.{ks[i]}", g_a, g_a_t, hs[i], hs_t[i])
        W = sd[pre + f"q_mlp.layers.{ks[i]}.weight"]
        g_h, g_h_t = g_a @ W, g_a_t @ W
    ln_w = sd[pre + "q_norm.weight"]
    G_[pre + "q_norm.weight"] = G_.get(pre + "q_norm.weight", 0) + (g_h_t * S["xh"] + g_h * S["xh_t"]).sum(0)
    G_[pre + "q_norm.bias"] = G_.get(pre + "q_norm.bias", 0) + g_h_t.sum(0)
    g_feat, g_feat_t = ln_bwd_dual(g_h, g_h_t, S["xh"], S["xh_t"], S["rstd"], S["rstd_t"], ln_w)
    Fh = S["X"].shape[2]
    gI, gA_, gS = g_feat[:, :Fh], g_feat[:, Fh:2 * Fh], g_feat[:, 2 * Fh:]
    gI_t, gA_t, gS_t = g_feat_t[:, :Fh], g_feat_t[:, Fh:2 * Fh], g_feat_t[:, 2 * Fh:]
    dq, dq_t = S["dq"], S["dq_t"]
    g_X = torch.cat([gI[:, None], dq[:, 1:4] * gA_[:, None], dq[:, 4:9] * gS[:, None]], 1)
    g_X_t = torch.cat([gI_t[:, None], dq_t[:, 1:4] * gA_[:, None] + dq[:, 1:4] * gA_t[:, None],
                       dq_t[:, 4:9] * gS[:, None] + dq[:, 4:9] * gS_t[:, None]], 1)
    return g_X, g_X_t


def _exp_cutoff_derivs(d, rc):
    """exp(-1 / (1 - x^2)) / e^-1 with x = clamp(d / rc, 0, 1 - 1e-6) (output_modules.py:323-332): value, d/dd, d2/dd2"""
    x = d / rc
    inside = x < 1.0 - 1e-6
    xc = x.clamp(0, 1.0 - 1e-6)
    u = 1.0 - xc * xc
    f = torch.exp(-1.0 / u) / 0.36787944117144233
    fx = f * (-2 * xc / u ** 2)
    fxx = f * (4 * xc ** 2 / u ** 4 - 2 / u ** 2 - 8 * xc ** 2 / u ** 3)
    z = torch.zeros_like(d)
    return f, torch.where(inside, fx / rc, z), torch.where(inside, fxx / rc ** 2, z)


def coulomb_dual(hp, ch, ch_t, pos, batch, v, qw, box=None):
    """damped pair Coulomb energy of all charge channels (output_modules.py:440-606), per atom, with its tangent along v, its
    adjoints wrt the charges (seed 1 on every atom's energy) and wrt the positions, and their tangents"""
    n = pos.shape[0]
    dt = pos.dtype
    cut = hp.get("coulomb_cutoff")
    if cut is None:
        i, j = torch.triu_indices(n, n, 1).unbind(0)
        m = batch[i] == batch[j]
        i, j = i[m], j[m]
        delta = pos[i] - pos[j]
        d = torch.linalg.norm(delta, dim=-1)
        g0, g1, g2 = 1.0 / d, -1.0 / d ** 2, 2.0 / d ** 3
    else:
        ei, d, delta = neighbor_pairs(pos, batch, 0.0, float(cut), box, loop=False, include_transpose=False)
        i, j = ei
        eps = float(hp.get("coulomb_epsilon_solvent", 78.3))
        k_rf = (1.0 / cut ** 3) * (eps - 1.0) / (2.0 * eps + 1.0)
        c_rf = (1.0 / cut) * (3.0 * eps) / (2.0 * eps + 1.0)
        g0, g1, g2 = 1.0 / d + k_rf * d ** 2 - c_rf, -1.0 / d ** 2 + 2 * k_rf * d, 2.0 / d ** 3 + 2 * k_rf
    # neighbor_pairs returns delta = pos[ei[0]] - pos[ei[1]] (+ image); orient the tangent the same way
    rhat = delta / d[:, None]
    delta_t = v[i] - v[j]
    d_t = (rhat * delta_t).sum(-1)
    rhat_t = (delta_t - rhat * d_t[:, None]) / d[:, None]
    f0, f1, f2 = _exp_cutoff_derivs(d, 4.6)
    h0 = COULOMB_FACTOR * (1.0 - f0) * g0
    h1 = COULOMB_FACTOR * (-f1 * g0 + (1.0 - f0) * g1)
    h2 = COULOMB_FACTOR * (-f2 * g0 - 2 * f1 * g1 + (1.0 - f0) * g2)
    wq = qw / qw.sum()
    S_ = (ch[i] * ch[j] * wq).sum(-1)
    S_t = ((ch_t[i] * ch[j] + ch[i] * ch_t[j]) * wq).sum(-1)
    e, e_t = h0 * S_, h1 * d_t * S_ + h0 * S_t
    e_atom = torch.zeros(n, dtype=dt).index_add(0, i, e).index_add(0, j, e)
    e_atom_t = torch.zeros(n, dtype=dt).index_add(0, i, e_t).index_add(0, j, e_t)
    # reverse, seed 1 on every atom's energy: the pair energy counts twice
    g_S, g_S_t = 2 * h0, 2 * h1 * d_t
    g_ch = torch.zeros_like(ch).index_add(0, i, g_S[:, None] * ch[j] * wq).index_add(0, j, g_S[:, None] * ch[i] * wq)
    g_ch_t = (torch.zeros_like(ch).index_add(0, i, (g_S_t[:, None] * ch[j] + g_S[:, None] * ch_t[j]) * wq)
              .index_add(0, j, (g_S_t[:, None] * ch[i] + g_S[:, None] * ch_t[i]) * wq))
    g_d, g_d_t = 2 * S_ * h1, 2 * (S_t * h1 + S_ * h2 * d_t)
    g_delta, g_delta_t = g_d[:, None] * rhat, g_d_t[:, None] * rhat + g_d[:, None] * rhat_t
    g_pos = torch.zeros(n, 3, dtype=dt).index_add(0, i, g_delta).index_add(0, j, -g_delta)
    Hv = torch.zeros(n, 3, dtype=dt).index_add(0, i, g_delta_t).index_add(0, j, -g_delta_t)
    return dict(e_atom=e_atom, e_atom_t=e_atom_t, g_ch=g_ch, g_ch_t=g_ch_t, g_pos=g_pos, Hv=Hv)


def force_term(sd, hp, z, pos, batch, v, box=None, q=None, ge=None):
    """returns dict(E, F, s, ent={TensorNet-style entry name: d s / d entry}, extra={state-dict key: gradient of the charge heads},
    Hv=d s / d pos, inter={name: tensor})."""
    R = "representation_model."
    Fh, L, K = hp["hidden_channels"], hp["num_layers"], hp["num_rbf"]
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    o3 = hp.get("equivariance_invariance_group", "O(3)") == "O(3)"
    dt = pos.dtype
    g = A.build_graph(pos, batch, lo, up, box)
    n, P = g["n"], g["P"]
    nmol = int(batch.max()) + 1
    qd = hp["q_dim"]
    Qatom = torch.zeros(n, dtype=dt) if q is None else q.to(dt)[batch]  # tensornet2.py:411-414
    kap4 = 1.0  # TensorNet2 has no charge factor in the group product / update (tensornet2.py:608-624)
    G_ = {}  # gradient entries that exist only here, by state-dict key
    pi, pj = g["pi"], g["pj"]
    rows, cols, ep, sg = g["rows"], g["cols"], g["epair"], g["esign"]
    NE = rows.shape[0]
    mm, tr_, compose, dec, compose_T, dec_T, quad, dquad = A.mm, A.tr_, A.compose, A.dec, A.compose_T, A.dec_T, A.quad, A.dquad
    ent, inter = {}, {}

    # ================= geometry and its tangent along v
    d, rhat = g["d"], g["rhat"]
    delta_t = v[pi] - v[pj]
    d_t = (rhat * delta_t).sum(-1)
    rhat_t = (delta_t - rhat * d_t[:, None]) / d[:, None]
    zero1 = torch.zeros(1, dtype=dt)
    dd, dd_t = torch.cat([d, zero1]), torch.cat([d_t, zero1])  # + self pair
    rh, rh_t = torch.cat([rhat, torch.zeros(1, 3, dtype=dt)]), torch.cat([rhat_t, torch.zeros(1, 3, dtype=dt)])
    phi, dphi, d2phi = rbf_derivs(dd, sd[R + "distance_expansion.means"], sd[R + "distance_expansion.betas"], lo, up)
    Cp, dCp, d2Cp = cutoff_derivs(dd, lo, up)
    phi_t, Cp_t = dphi * dd_t[:, None], dCp * dd_t
    re, re_t = sg[:, None] * rh[ep], sg[:, None] * rh_t[ep]
    inter.update(d_t=d_t, rhat_t=rhat_t, phi_t=phi_t, C_t=Cp_t)

    # ================= forward + tangent: embedding
    T = R + "tensor_embedding."
    Wdp = torch.cat([sd[T + f"distance_proj{k}.weight"] for k in (1, 2, 3)], 0)
    bdp = torch.cat([sd[T + f"distance_proj{k}.bias"] for k in (1, 2, 3)], 0)
    Q, Q_t = (phi @ Wdp.t() + bdp).view(P + 1, 3, Fh), (phi_t @ Wdp.t()).view(P + 1, 3, Fh)
    W2 = sd[T + "emb2.weight"]
    Utab = sd[T + "emb.weight"] @ W2[:, :Fh].t() + sd[T + "emb2.bias"]
    Vtab = sd[T + "emb.weight"] @ W2[:, Fh:].t()
    Zij = Utab[z[rows]] + Vtab[z[cols]]
    CZ, CZ_t = Cp[ep, None] * Zij, Cp_t[ep, None] * Zij
    Wd = CZ[:, None, :] * Q[ep]
    Wd_t = CZ_t[:, None, :] * Q[ep] + CZ[:, None, :] * Q_t[ep]
    r6, r6_t = r6_of(re), r6_dual(re, re_t)

    def scat(e_vals, width):
        return torch.zeros((n,) + tuple(width), dtype=dt).index_add(0, rows, e_vals)

    I0, I0_t = scat(Wd[:, 0], (Fh,)), scat(Wd_t[:, 0], (Fh,))
    vec = scat(Wd[:, 1, None, :] * re[:, :, None], (3, Fh))
    vec_t = scat(Wd_t[:, 1, None, :] * re[:, :, None] + Wd[:, 1, None, :] * re_t[:, :, None], (3, Fh))
    T6 = scat(Wd[:, 2, None, :] * r6[:, :, None], (6, Fh))
    T6_t = scat(Wd_t[:, 2, None, :] * r6[:, :, None] + Wd[:, 2, None, :] * r6_t[:, :, None], (6, Fh))
    u0, u0_t = u0_of(I0, vec, T6), u0_of(I0_t, vec_t, T6_t)
    s0n, s0n_t = quad(u0), (dquad(u0) * u0_t).sum(1)
    ln0_w, ln0_b = sd[T + "init_norm.weight"], sd[T + "init_norm.bias"]
    ln0, xh0, rstd0, ln0_t, xh0_t, rstd0_t = ln_dual(s0n, s0n_t, ln0_w, ln0_b)
    L1, bL1 = sd[T + "linears_scalar.0.weight"], sd[T + "linears_scalar.0.bias"]
    L2, bL2 = sd[T + "linears_scalar.1.weight"], sd[T + "linears_scalar.1.bias"]
    a1, a1_t = ln0 @ L1.t() + bL1, ln0_t @ L1.t()
    h1, h1_t = Fn.silu(a1), silu_d1(a1) * a1_t
    a2, a2_t = h1 @ L2.t() + bL2, h1_t @ L2.t()
    gates, gates_t = Fn.silu(a2).view(n, 3, Fh), (silu_d1(a2) * a2_t).view(n, 3, Fh)
    Ue = [sd[T + f"linears_tensor.{k}.weight"] for k in range(3)]
    UX, UX_t = tlin(u0, Ue), tlin(u0_t, Ue)
    X = UX * gates[:, TYPE_OF, :]
    X_t = UX_t * gates[:, TYPE_OF, :] + UX * gates_t[:, TYPE_OF, :]
    inter.update(Q_t=Q_t, u0_t=u0_t, s0n_t=s0n_t, ln0_t=ln0_t, a1_t=a1_t, a2_t=a2_t, UX_t=UX_t, X0_t=X_t)
    cps = [charge_predict_dual(sd, R + "charge_predict_0.", X, X_t, batch, Qatom, nmol, qd)]

    # ================= forward + tangent: interaction layers
    cache = []
    for l in range(L):
        Lp = R + f"layers.{l}."
        M1, b1 = sd[Lp + "linears_scalar.0.weight"], sd[Lp + "linears_scalar.0.bias"]
        M2, b2 = sd[Lp + "linears_scalar.1.weight"], sd[Lp + "linears_scalar.1.bias"]
        M3, b3 = sd[Lp + "linears_scalar.2.weight"], sd[Lp + "linears_scalar.2.bias"]
        V = [sd[Lp + f"linears_tensor.{k}.weight"] for k in range(6)]
        ch, ch_t = cps[-1]["ch"], cps[-1]["ch_t"]
        ein = torch.cat([phi[ep], ch[rows], ch[cols]], -1)  # [phi(d), c_i, c_j] per DIRECTED edge (tensornet2.py:548-566)
        ein_t = torch.cat([phi_t[ep], ch_t[rows], ch_t[cols]], -1)
        e1, e1_t = ein @ M1.t() + b1, ein_t @ M1.t()
        he1, he1_t = Fn.silu(e1), silu_d1(e1) * e1_t
        e2, e2_t = he1 @ M2.t() + b2, he1_t @ M2.t()
        he2, he2_t = Fn.silu(e2), silu_d1(e2) * e2_t
        e3, e3_t = he2 @ M3.t() + b3, he2_t @ M3.t()
        f3 = Fn.silu(e3)
        w = (f3 * Cp[ep, None]).view(NE, 3, Fh)
        w_t = (silu_d1(e3) * e3_t * Cp[ep, None] + f3 * Cp_t[ep, None]).view(NE, 3, Fh)
        s, s_t = quad(X), (dquad(X) * X_t).sum(1)
        sp1 = (s + 1)[:, None, :]
        Xh = X / sp1
        Xh_t = X_t / sp1 - X * s_t[:, None, :] / sp1 ** 2
        Pn, Pn_t = tlin(Xh, V[0:3]), tlin(Xh_t, V[0:3])
        Mi = edge_gather(Pn, w, rows, cols)
        Mi_t = edge_gather(Pn_t, w, rows, cols) + edge_gather(Pn, w_t, rows, cols)
        Y, Y_t, Mf, Mf_t = compose(Pn), compose(Pn_t), compose(Mi), compose(Mi_t)
        if o3:
            Cm = kap4 * (mm(Y, Mf) + mm(Mf, Y))
            Cm_t = kap4 * (mm(Y_t, Mf) + mm(Y, Mf_t) + mm(Mf_t, Y) + mm(Mf, Y_t))
        else:
            Cm, Cm_t = 2 * mm(Y, Mf), 2 * (mm(Y_t, Mf) + mm(Y, Mf_t))
        uC, uC_t = dec(Cm), dec(Cm_t)
        t, t_t = (Cm * Cm).sum((1, 2)), 2 * (Cm * Cm_t).sum((1, 2))
        tp1 = (t + 1)[:, None, :]
        Ch = uC / tp1
        Ch_t = uC_t / tp1 - uC * t_t[:, None, :] / tp1 ** 2
        D, D_t = tlin(Ch, V[3:6]), tlin(Ch_t, V[3:6])
        dX, dX_t = compose(D), compose(D_t)
        Xn = Xh + D + dec(kap4 * mm(dX, dX))
        Xn_t = Xh_t + D_t + dec(kap4 * (mm(dX_t, dX) + mm(dX, dX_t)))
        cache.append(dict(X=X, X_t=X_t, s=s, s_t=s_t, Xh=Xh, Xh_t=Xh_t, Pn=Pn, Pn_t=Pn_t, Y=Y, Y_t=Y_t, Mf=Mf, Mf_t=Mf_t, Cm=Cm, Cm_t=Cm_t,
                          uC=uC, uC_t=uC_t, t=t, t_t=t_t, Ch=Ch, Ch_t=Ch_t, dX=dX, dX_t=dX_t, e1=e1, e1_t=e1_t, e2=e2, e2_t=e2_t, e3=e3,
                          e3_t=e3_t, he1=he1, he1_t=he1_t, he2=he2, he2_t=he2_t, w=w, w_t=w_t, V=V, M1=M1, M2=M2, M3=M3, ein=ein, ein_t=ein_t))
        inter.update({f"l{l}.w_t": w_t, f"l{l}.Xh_t": Xh_t, f"l{l}.Pn_t": Pn_t, f"l{l}.Mi_t": Mi_t, f"l{l}.Ch_t": Ch_t, f"l{l}.D_t": D_t,
                      f"l{l}.Xn_t": Xn_t})
        X, X_t = Xn, Xn_t
        cps.append(charge_predict_dual(sd, R + f"charge_predicts.{l}.", X, X_t, batch, Qatom, nmol, qd))

    # ================= forward + tangent: readout, head
    dq, dq_t = dquad(X), dquad(X_t)
    I = X[:, 0]
    feat = torch.cat([3 * I * I, 2 * (X[:, 1:4] ** 2).sum(1),
                      X[:, 4] ** 2 + X[:, 7] ** 2 + (X[:, 4] + X[:, 7]) ** 2 + 2 * (X[:, 5] ** 2 + X[:, 6] ** 2 + X[:, 8] ** 2)], -1)
    feat_t = tsum(dq * X_t).reshape(n, 3 * Fh)
    lnr_w, lnr_b = sd[R + "out_norm.weight"], sd[R + "out_norm.bias"]
    lnr, xhr, rstdr, lnr_t, xhr_t, rstdr_t = ln_dual(feat, feat_t, lnr_w, lnr_b)
    Lin, bLin = sd[R + "linear.weight"], sd[R + "linear.bias"]
    al, al_t = lnr @ Lin.t() + bLin, lnr_t @ Lin.t()
    x, x_t = Fn.silu(al), silu_d1(al) * al_t
    O = "output_model.output_network.layers."
    O1, bO1, O2, bO2 = sd[O + "0.weight"], sd[O + "0.bias"], sd[O + "2.weight"], sd[O + "2.bias"]
    ao, ao_t = x @ O1.t() + bO1, x_t @ O1.t()
    std = sd.get("std", torch.ones((), dtype=dt)).to(dt)
    charges, charges_t = torch.cat([c_["ch"] for c_ in cps], -1), torch.cat([c_["ch_t"] for c_ in cps], -1)
    cou = coulomb_dual(hp, charges, charges_t, pos, batch, v, sd["output_model.qweights"].to(dt), box)
    ea = (Fn.silu(ao) @ O2.t() + bO2 + cou["e_atom"][:, None]) * std
    ea_t = ((silu_d1(ao) * ao_t) @ O2.t() + cou["e_atom_t"][:, None]) * std
    E = torch.zeros(nmol, 1, dtype=dt).index_add(0, batch, ea) + sd.get("mean", torch.zeros((), dtype=dt)).to(dt)
    s_val = ea_t.sum()
    inter.update(feat_t=feat_t, lnr_t=lnr_t, al_t=al_t, ao_t=ao_t, ea_t=ea_t)

    # ================= reverse pass (seed 1 on every atom's energy) and its tangent
    g_ao, g_ao_t = (std * O2).expand(n, -1) * silu_d1(ao), (std * O2).expand(n, -1) * silu_d2(ao) * ao_t
    ent["O2"] = std * (silu_d1(ao) * ao_t).sum(0, keepdim=True)
    ent["bO2"] = torch.zeros(1, dtype=dt)
    cou_g_ch_t, cou_Hv = cou["g_ch_t"], cou["Hv"]
    if ge is not None:  # one-pass training: gradient of S = s - sum_m ge_m E_m (oracle/tensornet_second_order.py force_term); the
        # Coulomb pairs lie inside a molecule, so its adjoints take the factor of their atom's molecule
        wa = ge.to(dt).reshape(-1)[batch][:, None]
        g_ao_t = g_ao_t - wa * g_ao
        ent["O2"] = ent["O2"] - std * (wa * Fn.silu(ao)).sum(0, keepdim=True)
        ent["bO2"] = ent["bO2"] - std * wa.sum()
        cou_g_ch_t, cou_Hv = cou["g_ch_t"] - wa * cou["g_ch"], cou["Hv"] - wa * cou["g_pos"]
        s_val = s_val - (ge.to(dt).reshape(-1) * E.reshape(-1)).sum()
    ent["O1"], ent["bO1"] = g_ao_t.t() @ x + g_ao.t() @ x_t, g_ao_t.sum(0)
    g_x, g_x_t = g_ao @ O1, g_ao_t @ O1
    g_al, g_al_t = g_x * silu_d1(al), g_x_t * silu_d1(al) + g_x * silu_d2(al) * al_t
    ent["Lin"], ent["bLin"] = g_al_t.t() @ lnr + g_al.t() @ lnr_t, g_al_t.sum(0)
    g_ln, g_ln_t = g_al @ Lin, g_al_t @ Lin
    ent["lnr_w"], ent["lnr_b"] = (g_ln_t * xhr + g_ln * xhr_t).sum(0), g_ln_t.sum(0)
    g_feat, g_feat_t = ln_bwd_dual(g_ln, g_ln_t, xhr, xhr_t, rstdr, rstdr_t, lnr_w)
    gf3, gf3_t = g_feat.view(n, 3, Fh)[:, TYPE_OF, :], g_feat_t.view(n, 3, Fh)[:, TYPE_OF, :]
    G, G_t = dq * gf3, dq_t * gf3 + dq * gf3_t
    inter.update(g_ao_t=g_ao_t, g_al_t=g_al_t, g_feat_t=g_feat_t, G_top_t=G_t)
    # adjoints of the charge channels: from the Coulomb term (all sets at once) and, below, from the edge MLP of the layer they feed
    g_chs = [std * cou["g_ch"][:, k * qd:(k + 1) * qd].clone() for k in range(L + 1)]
    g_chs_t = [std * cou_g_ch_t[:, k * qd:(k + 1) * qd].clone() for k in range(L + 1)]
    # charge head on the final X: adjoint into the residual stream
    gx, gx_t = charge_predict_bwd_dual(sd, R + (f"charge_predicts.{L - 1}." if L > 0 else "charge_predict_0."), cps[L], g_chs[L], g_chs_t[L],
                                       batch, nmol, qd, G_)
    G, G_t = G + gx, G_t + gx_t

    g_phi, g_phi_t = torch.zeros(P + 1, K, dtype=dt), torch.zeros(P + 1, K, dtype=dt)
    g_C, g_C_t = torch.zeros(P + 1, dtype=dt), torch.zeros(P + 1, dtype=dt)
    for l in reversed(range(L)):
        c = cache[l]
        V = c["V"]
        dX, dX_t = c["dX"], c["dX_t"]
        Gf, Gf_t = dec_T(G), dec_T(G_t)
        g_dXf = Gf + kap4 * (mm(Gf, tr_(dX)) + mm(tr_(dX), Gf))
        g_dXf_t = Gf_t + kap4 * (mm(Gf_t, tr_(dX)) + mm(tr_(dX), Gf_t) + mm(Gf, tr_(dX_t)) + mm(tr_(dX_t), Gf))
        g_D, g_D_t = compose_T(g_dXf), compose_T(g_dXf_t)
        for k, (a_, b_) in enumerate(zip(tlin_grad(g_D_t, c["Ch"]), tlin_grad(g_D, c["Ch_t"]))):
            ent[f"l{l}.Vb{k}"] = a_ + b_
        g_Ch, g_Ch_t = tlin(g_D, V[3:6], True), tlin(g_D_t, V[3:6], True)
        tp1 = (c["t"] + 1)[:, None, :]
        t_t3 = c["t_t"][:, None, :]
        g_uC, g_uC_t = g_Ch / tp1, g_Ch_t / tp1 - g_Ch * t_t3 / tp1 ** 2
        a_ = (g_Ch * c["uC"]).sum(1)
        a_t = (g_Ch_t * c["uC"] + g_Ch * c["uC_t"]).sum(1)
        g_t = -a_ / (c["t"] + 1) ** 2
        g_t_t = -a_t / (c["t"] + 1) ** 2 + 2 * a_ * c["t_t"] / (c["t"] + 1) ** 3
        g_Cm = dec_T(g_uC) + 2 * c["Cm"] * g_t[:, None, None, :]
        g_Cm_t = dec_T(g_uC_t) + 2 * (c["Cm_t"] * g_t[:, None, None, :] + c["Cm"] * g_t_t[:, None, None, :])
        Y, Y_t, Mf, Mf_t = c["Y"], c["Y_t"], c["Mf"], c["Mf_t"]
        if o3:
            g_Y = kap4 * (mm(g_Cm, tr_(Mf)) + mm(tr_(Mf), g_Cm))
            g_Y_t = kap4 * (mm(g_Cm_t, tr_(Mf)) + mm(tr_(Mf), g_Cm_t) + mm(g_Cm, tr_(Mf_t)) + mm(tr_(Mf_t), g_Cm))
            g_Mf = kap4 * (mm(tr_(Y), g_Cm) + mm(g_Cm, tr_(Y)))
            g_Mf_t = kap4 * (mm(tr_(Y_t), g_Cm) + mm(g_Cm, tr_(Y_t)) + mm(tr_(Y), g_Cm_t) + mm(g_Cm_t, tr_(Y)))
        else:
            g_Y, g_Y_t = 2 * mm(g_Cm, tr_(Mf)), 2 * (mm(g_Cm_t, tr_(Mf)) + mm(g_Cm, tr_(Mf_t)))
            g_Mf, g_Mf_t = 2 * mm(tr_(Y), g_Cm), 2 * (mm(tr_(Y_t), g_Cm) + mm(tr_(Y), g_Cm_t))
        g_Mi, g_Mi_t = compose_T(g_Mf), compose_T(g_Mf_t)
        g_Pn = compose_T(g_Y) + edge_gather_T(g_Mi, c["w"], rows, cols, n)
        g_Pn_t = compose_T(g_Y_t) + edge_gather_T(g_Mi_t, c["w"], rows, cols, n) + edge_gather_T(g_Mi, c["w_t"], rows, cols, n)
        # per-pair weight gradient [P + 1, 3F]; the self pair (row P) collects every atom's self edge
        Pn, Pn_t = c["Pn"], c["Pn_t"]

        def edge_rows(gm, pn):  # per DIRECTED edge (the weights differ by direction: they see c_i, c_j)
            return tsum(gm[rows] * pn[cols]).reshape(NE, 3 * Fh)

        g_w, g_w_t = edge_rows(g_Mi, Pn), edge_rows(g_Mi_t, Pn) + edge_rows(g_Mi, Pn_t)
        e1, e2, e3, e1_t, e2_t, e3_t = c["e1"], c["e2"], c["e3"], c["e1_t"], c["e2_t"], c["e3_t"]
        f3 = Fn.silu(e3)
        g_C.index_add_(0, ep, (g_w * f3).sum(-1))
        g_C_t.index_add_(0, ep, (g_w_t * f3 + g_w * silu_d1(e3) * e3_t).sum(-1))
        C1, C1_t = Cp[ep, None], Cp_t[ep, None]
        g_e3 = g_w * C1 * silu_d1(e3)
        g_e3_t = g_w_t * C1 * silu_d1(e3) + g_w * C1_t * silu_d1(e3) + g_w * C1 * silu_d2(e3) * e3_t
        ent[f"l{l}.M2"], ent[f"l{l}.b2"] = g_e3_t.t() @ c["he2"] + g_e3.t() @ c["he2_t"], g_e3_t.sum(0)
        g_he2, g_he2_t = g_e3 @ c["M3"], g_e3_t @ c["M3"]
        g_e2, g_e2_t = g_he2 * silu_d1(e2), g_he2_t * silu_d1(e2) + g_he2 * silu_d2(e2) * e2_t
        ent[f"l{l}.M1"], ent[f"l{l}.b1"] = g_e2_t.t() @ c["he1"] + g_e2.t() @ c["he1_t"], g_e2_t.sum(0)
        g_he1, g_he1_t = g_e2 @ c["M2"], g_e2_t @ c["M2"]
        g_e1, g_e1_t = g_he1 * silu_d1(e1), g_he1_t * silu_d1(e1) + g_he1 * silu_d2(e1) * e1_t
        ent[f"l{l}.M0"], ent[f"l{l}.b0"] = g_e1_t.t() @ c["ein"] + g_e1.t() @ c["ein_t"], g_e1_t.sum(0)
        g_in, g_in_t = g_e1 @ c["M1"], g_e1_t @ c["M1"]  # [E, K + 2 q_dim]
        g_phi.index_add_(0, ep, g_in[:, :K])
        g_phi_t.index_add_(0, ep, g_in_t[:, :K])
        g_chs[l] = g_chs[l].index_add(0, rows, g_in[:, K:K + qd]).index_add(0, cols, g_in[:, K + qd:])
        g_chs_t[l] = g_chs_t[l].index_add(0, rows, g_in_t[:, K:K + qd]).index_add(0, cols, g_in_t[:, K + qd:])
        for k, (a2_, b2_) in enumerate(zip(tlin_grad(g_Pn_t, c["Xh"]), tlin_grad(g_Pn, c["Xh_t"]))):
            ent[f"l{l}.Va{k}"] = a2_ + b2_
        g_Xh, g_Xh_t = G + tlin(g_Pn, V[0:3], True), G_t + tlin(g_Pn_t, V[0:3], True)
        Xl, Xl_t, s, s_t = c["X"], c["X_t"], c["s"], c["s_t"]
        sp1 = (s + 1)[:, None, :]
        b_ = (g_Xh * Xl).sum(1)
        b_t = (g_Xh_t * Xl + g_Xh * Xl_t).sum(1)
        g_s = -b_ / (s + 1) ** 2
        g_s_t = -b_t / (s + 1) ** 2 + 2 * b_ * s_t / (s + 1) ** 3
        G_new = g_Xh / sp1 + dquad(Xl) * g_s[:, None, :]
        G_t = g_Xh_t / sp1 - g_Xh * s_t[:, None, :] / sp1 ** 2 + dquad(Xl_t) * g_s[:, None, :] + dquad(Xl) * g_s_t[:, None, :]
        G = G_new
        # the charge head on this layer's input X_l: its channels went into the Coulomb term and into this layer's edge MLP
        gx, gx_t = charge_predict_bwd_dual(sd, R + (f"charge_predicts.{l - 1}." if l > 0 else "charge_predict_0."), cps[l], g_chs[l],
                                           g_chs_t[l], batch, nmol, qd, G_)
        G, G_t = G + gx, G_t + gx_t
        inter.update({f"l{l}.g_Ch_t": g_Ch_t, f"l{l}.g_Mi_t": g_Mi_t, f"l{l}.g_Pn_t": g_Pn_t, f"l{l}.g_w_t": g_w_t, f"l{l}.g_e3_t": g_e3_t,
                      f"l{l}.g_e1_t": g_e1_t, f"l{l}.G_t": G_t})

    # ---- embedding adjoint and its tangent
    gT, gT_t = gates[:, TYPE_OF, :], gates_t[:, TYPE_OF, :]
    g_UX, g_UX_t = G * gT, G_t * gT + G * gT_t
    g_gates, g_gates_t = tsum(G * UX).reshape(n, 3 * Fh), tsum(G_t * UX + G * UX_t).reshape(n, 3 * Fh)
    g_a2, g_a2_t = g_gates * silu_d1(a2), g_gates_t * silu_d1(a2) + g_gates * silu_d2(a2) * a2_t
    ent["L2"], ent["bL2"] = g_a2_t.t() @ h1 + g_a2.t() @ h1_t, g_a2_t.sum(0)
    g_h1, g_h1_t = g_a2 @ L2, g_a2_t @ L2
    g_a1, g_a1_t = g_h1 * silu_d1(a1), g_h1_t * silu_d1(a1) + g_h1 * silu_d2(a1) * a1_t
    ent["L1"], ent["bL1"] = g_a1_t.t() @ ln0 + g_a1.t() @ ln0_t, g_a1_t.sum(0)
    g_ln0, g_ln0_t = g_a1 @ L1, g_a1_t @ L1
    ent["ln0_w"], ent["ln0_b"] = (g_ln0_t * xh0 + g_ln0 * xh0_t).sum(0), g_ln0_t.sum(0)
    g_s0n, g_s0n_t = ln_bwd_dual(g_ln0, g_ln0_t, xh0, xh0_t, rstd0, rstd0_t, ln0_w)
    for k, (a_, b_) in enumerate(zip(tlin_grad(g_UX_t, u0), tlin_grad(g_UX, u0_t))):
        ent[f"Ue{k}"] = a_ + b_
    g_u0 = tlin(g_UX, Ue, True) + dquad(u0) * g_s0n[:, None, :]
    g_u0_t = tlin(g_UX_t, Ue, True) + dquad(u0_t) * g_s0n[:, None, :] + dquad(u0) * g_s0n_t[:, None, :]
    gA, gA_t = gA_of(g_u0), gA_of(g_u0_t)  # [N, 10, F]: I0, vec, T6
    inter.update(g_a2_t=g_a2_t, g_a1_t=g_a1_t, g_s0n_t=g_s0n_t, gA_t=gA_t)
    # per directed edge
    gI0, gv, gT6 = gA[:, 0], gA[:, 1:4], gA[:, 4:10]
    gI0_t, gv_t, gT6_t = gA_t[:, 0], gA_t[:, 1:4], gA_t[:, 4:10]
    gWd = torch.stack([gI0[rows], (gv[rows] * re[:, :, None]).sum(1), (gT6[rows] * r6[:, :, None]).sum(1)], 1)  # [E, 3, F]
    gWd_t = torch.stack([gI0_t[rows], (gv_t[rows] * re[:, :, None] + gv[rows] * re_t[:, :, None]).sum(1),
                         (gT6_t[rows] * r6[:, :, None] + gT6[rows] * r6_t[:, :, None]).sum(1)], 1)
    g_Q = torch.zeros(P + 1, 3, Fh, dtype=dt).index_add(0, ep, gWd * CZ[:, None, :])
    g_Q_t = torch.zeros(P + 1, 3, Fh, dtype=dt).index_add(0, ep, gWd_t * CZ[:, None, :] + gWd * CZ_t[:, None, :])
    g_Q2, g_Q2_t = g_Q.reshape(P + 1, 3 * Fh), g_Q_t.reshape(P + 1, 3 * Fh)
    ent["Wdp"], ent["bdp"] = g_Q2_t.t() @ phi + g_Q2.t() @ phi_t, g_Q2_t.sum(0)
    Qe, Qe_t, Ce, Ce_t = Q[ep], Q_t[ep], Cp[ep, None, None], Cp_t[ep, None, None]
    g_Z_t = (gWd_t * Ce * Qe + gWd * Ce_t * Qe + gWd * Ce * Qe_t).sum(1)  # [E, F]
    nz = sd[T + "emb.weight"].shape[0]
    ent["Utab"] = torch.zeros(nz, Fh, dtype=dt).index_add(0, z[rows], g_Z_t)
    ent["Vtab"] = torch.zeros(nz, Fh, dtype=dt).index_add(0, z[cols], g_Z_t)
    inter.update(g_Q_t=g_Q_t, g_Z_t=g_Z_t)

    # ---- geometry: H v  (position gradient of s)
    g_C.index_add_(0, ep, (gWd * Zij[:, None, :] * Qe).sum((1, 2)))
    g_C_t.index_add_(0, ep, ((gWd_t * Qe + gWd * Qe_t) * Zij[:, None, :]).sum((1, 2)))
    g_phi += g_Q2 @ Wdp
    g_phi_t += g_Q2_t @ Wdp
    gTe, gTe_t = gT6[rows], gT6_t[rows]

    def dq_rows(gt, r):
        return torch.stack([2 * gt[:, 0] * r[:, 0, None] + gt[:, 1] * r[:, 1, None] + gt[:, 2] * r[:, 2, None],
                            gt[:, 1] * r[:, 0, None] + 2 * gt[:, 3] * r[:, 1, None] + gt[:, 4] * r[:, 2, None],
                            gt[:, 2] * r[:, 0, None] + gt[:, 4] * r[:, 1, None] + 2 * gt[:, 5] * r[:, 2, None]], 1)  # [E, 3, F]

    dqr, dqr_t = dq_rows(gTe, re), dq_rows(gTe_t, re) + dq_rows(gTe, re_t)
    g_re = (gv[rows] * Wd[:, 1, None, :]).sum(-1) + (dqr * Wd[:, 2, None, :]).sum(-1)
    g_re_t = ((gv_t[rows] * Wd[:, 1, None, :] + gv[rows] * Wd_t[:, 1, None, :]).sum(-1)
              + (dqr_t * Wd[:, 2, None, :] + dqr * Wd_t[:, 2, None, :]).sum(-1))
    g_rh = torch.zeros(P + 1, 3, dtype=dt).index_add(0, ep, sg[:, None] * g_re)[:P]
    g_rh_t = torch.zeros(P + 1, 3, dtype=dt).index_add(0, ep, sg[:, None] * g_re_t)[:P]
    g_d = g_C[:P] * dCp[:P] + (g_phi[:P] * dphi[:P]).sum(-1)
    g_d_t = (g_C_t[:P] * dCp[:P] + g_C[:P] * d2Cp[:P] * d_t + (g_phi_t[:P] * dphi[:P]).sum(-1)
             + (g_phi[:P] * d2phi[:P]).sum(-1) * d_t)
    a_ = (g_rh * rhat).sum(-1, keepdim=True)
    a_t = (g_rh_t * rhat + g_rh * rhat_t).sum(-1, keepdim=True)
    term1 = (g_rh - a_ * rhat) / d[:, None]
    term1_t = (g_rh_t - a_t * rhat - a_ * rhat_t) / d[:, None] - term1 * (d_t / d)[:, None]
    g_delta = term1 + g_d[:, None] * rhat
    g_delta_t = term1_t + g_d_t[:, None] * rhat + g_d[:, None] * rhat_t
    g_pos = torch.zeros(n, 3, dtype=dt).index_add(0, pi, g_delta).index_add(0, pj, -g_delta) + std * cou["g_pos"]
    Hv = torch.zeros(n, 3, dtype=dt).index_add(0, pi, g_delta_t).index_add(0, pj, -g_delta_t) + std * cou_Hv
    inter.update(g_d_t=g_d_t, g_rh_t=g_rh_t)
    return dict(E=E, F=-g_pos, s=s_val, ent=ent, extra=G_, Hv=Hv, inter=inter)




def state_dict_grads(out, sd, hp):
    """entries of force_term -> gradients by state-dict key (TensorNet's chain rule for the species tables; the first edge layer is
    one [F, K + 2 q_dim] matrix here; the charge heads are already keyed)"""
    from oracle.tensornet_second_order import state_dict_grads as tn_grads

    g = tn_grads(out["ent"], sd, hp)
    g.update(out["extra"])
    return g
