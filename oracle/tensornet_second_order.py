"""TEST INFRASTRUCTURE ONLY -- CPU specification of the SECOND-ORDER pass of force-matching training on TensorNet:

    s(theta) = v . d(sum_m E_m)/d pos            (v = d loss / d F, so  d loss / d theta |_forces = - d s / d theta)

The reference gets d s / d theta by differentiating its autograd graph twice (torchmdnet/models/model.py:618-628,
create_graph=self.training; the *_bwd_bwd kernels of warp_ops/).  Here it is written out by hand, with no autograd, as the
forward-mode tangent (direction v in the positions) of the hand-derived E + F program of oracle/tensornet_adjoint.py:
every intermediate a of the forward pass gets a tangent a_t, every adjoint g_a of the reverse pass gets a tangent g_a_t,
and where a weight gradient is  g_W = g_y^T x  its tangent is  g_y_t^T x + g_y^T x_t  -- which is d s / d W.
Layout, graph and naming follow tensornet_adjoint.py; the returned gradient entries use the engine's names
(csrc/tn_api.hip train_layout: Wdp, bdp, Utab, Vtab, Ue0.., L1, .., l0.M0, .., lnr_w, .., O2, bO2), so the same dictionary
checks the HIP pass entry by entry.  tests/test_oracle.py pins this file to autograd-of-autograd over tensornet_torch.py.
"""
import math

import torch
import torch.nn.functional as Fn

from . import tensornet_adjoint as A

TYPE_OF = A.TYPE_OF
C0, NC = (0, 1, 4), (1, 3, 5)  # first component / component count of the irreducible types I, A, S


def silu_d1(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def silu_d2(x):
    s = torch.sigmoid(x)
    return s * (1 - s) * (2 + x * (1 - 2 * s))


def cutoff_derivs(d, lo, up):
    """CosineCutoff (models/utils.py:506-528): value, first and second derivative."""
    if lo > 0:
        k = 2 * math.pi / (up - lo)
        arg = math.pi * (2 * (d - lo) / (up - lo) + 1.0)
        m = ((d < up) & (d > lo)).to(d.dtype)
        return 0.5 * (torch.cos(arg) + 1.0) * m, -0.5 * torch.sin(arg) * k * m, -0.5 * torch.cos(arg) * k * k * m
    k = math.pi / up
    m = (d < up).to(d.dtype)
    return 0.5 * (torch.cos(d * k) + 1.0) * m, -0.5 * torch.sin(d * k) * k * m, -0.5 * torch.cos(d * k) * k * k * m


def rbf_derivs(d, means, betas, lo, up):
    """ExpNormalSmearing (models/utils.py:402-407): phi_k, d phi_k / dd, d2 phi_k / dd2."""
    alpha = 5.0 / (up - lo)
    c0, c1, c2 = cutoff_derivs(d, 0.0, up)
    c0, c1, c2 = c0[:, None], c1[:, None], c2[:, None]
    u = torch.exp(-alpha * (d - lo))[:, None]
    u1, u2 = -alpha * u, alpha * alpha * u
    g = torch.exp(-betas * (u - means) ** 2)
    h = -2 * betas * (u - means) * u1          # g' = g h
    h1 = -2 * betas * (u1 * u1 + (u - means) * u2)
    g1, g2 = g * h, g * (h * h + h1)
    return c0 * g, c1 * g + c0 * g1, c2 * g + 2 * c1 * g1 + c0 * g2


def tsum(x):
    """[*, 9, F] -> [*, 3, F]: sum over the components of each irreducible type."""
    return torch.stack([x[:, 0], x[:, 1:4].sum(1), x[:, 4:9].sum(1)], 1)


def tlin(u, Ws, transpose=False):
    return A.tensor_linear(u, Ws, transpose)


def tlin_grad(g_out, inp):
    """d/dW_k of  out_c = in_c W_k^T  summed over the atoms and the components of type k: [3][F_out, F_in]."""
    return [sum(g_out[:, c].t() @ inp[:, c] for c in range(C0[k], C0[k] + NC[k])) for k in range(3)]


def ln_dual(x, x_t, w, b, eps=1e-5):
    y, xh, rstd = A.layer_norm_fwd(x, w, b, eps)
    m = (xh * x_t).mean(-1, keepdim=True)
    xh_t = rstd * (x_t - x_t.mean(-1, keepdim=True) - xh * m)
    return y, xh, rstd, xh_t * w, xh_t, -rstd * rstd * m


def ln_bwd_dual(g, g_t, xh, xh_t, rstd, rstd_t, w):
    gw, gw_t = g * w, g_t * w
    c1, c2 = gw.mean(-1, keepdim=True), (gw * xh).mean(-1, keepdim=True)
    c1_t, c2_t = gw_t.mean(-1, keepdim=True), (gw_t * xh + gw * xh_t).mean(-1, keepdim=True)
    core = gw - c1 - xh * c2
    return core * rstd, (gw_t - c1_t - xh_t * c2 - xh * c2_t) * rstd + core * rstd_t


def u0_of(I0, vec, T6):
    trT = (T6[:, 0] + T6[:, 3] + T6[:, 5]) / 3
    return torch.stack([I0, vec[:, 0], vec[:, 1], vec[:, 2], T6[:, 0] - trT, T6[:, 1], T6[:, 2], T6[:, 3] - trT, T6[:, 4]], 1)


def gA_of(g_u0):
    """adjoint of u0_of: gradient wrt (I0, vec[3], T00, T01, T02, T11, T12, T22), [N, 10, F]."""
    gs0, gs3 = g_u0[:, 4], g_u0[:, 7]
    third = (gs0 + gs3) / 3
    return torch.stack([g_u0[:, 0], g_u0[:, 1], g_u0[:, 2], g_u0[:, 3], gs0 - third, g_u0[:, 5], g_u0[:, 6], gs3 - third, g_u0[:, 8],
                        -third], 1)


def r6_of(re):
    return torch.stack([re[:, 0] ** 2, re[:, 0] * re[:, 1], re[:, 0] * re[:, 2], re[:, 1] ** 2, re[:, 1] * re[:, 2], re[:, 2] ** 2], 1)


def r6_dual(re, re_t):
    x, y, z = re.unbind(1)
    a, b, c = re_t.unbind(1)
    return torch.stack([2 * x * a, a * y + x * b, a * z + x * c, 2 * y * b, b * z + y * c, 2 * z * c], 1)


def force_term(sd, hp, z, pos, batch, v, box=None, q=None, ge=None):
    """returns dict(E, F, s, ent={engine entry name: d s / d entry}, Hv=d s / d pos, inter={name: tensor}).
    ge [n_mol] (one-pass training): everything is the gradient of  S = s - sum_m ge_m E_m  instead - the tangent of the adjoint
    h = g_t - (adjoint of sum_m ge_m E_m) obeys the recursion of g_t, so only its seed at the head changes: with v = d loss / d F and
    ge = d loss / d E the whole parameter gradient of loss(E, F) is  - ent  and the position gradient  - Hv."""
    R = "representation_model."
    Fh, L, K = hp["hidden_channels"], hp["num_layers"], hp["num_rbf"]
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    o3 = hp.get("equivariance_invariance_group", "O(3)") == "O(3)"
    dt = pos.dtype
    g = A.build_graph(pos, batch, lo, up, box)
    n, P = g["n"], g["P"]
    nmol = int(batch.max()) + 1
    kap = torch.ones(n, dtype=dt) if q is None else 1 + 0.1 * q.to(dt)[batch]
    kap4 = kap[:, None, None, None]
    pi, pj = g["pi"], g["pj"]
    rows, cols, ep, sg = g["rows"], g["cols"], g["epair"], g["esign"]
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

    # ================= forward + tangent: interaction layers
    cache = []
    for l in range(L):
        Lp = R + f"layers.{l}."
        M1, b1 = sd[Lp + "linears_scalar.0.weight"], sd[Lp + "linears_scalar.0.bias"]
        M2, b2 = sd[Lp + "linears_scalar.1.weight"], sd[Lp + "linears_scalar.1.bias"]
        M3, b3 = sd[Lp + "linears_scalar.2.weight"], sd[Lp + "linears_scalar.2.bias"]
        V = [sd[Lp + f"linears_tensor.{k}.weight"] for k in range(6)]
        e1, e1_t = phi @ M1.t() + b1, phi_t @ M1.t()
        he1, he1_t = Fn.silu(e1), silu_d1(e1) * e1_t
        e2, e2_t = he1 @ M2.t() + b2, he1_t @ M2.t()
        he2, he2_t = Fn.silu(e2), silu_d1(e2) * e2_t
        e3, e3_t = he2 @ M3.t() + b3, he2_t @ M3.t()
        f3 = Fn.silu(e3)
        w = (f3 * Cp[:, None]).view(P + 1, 3, Fh)
        w_t = (silu_d1(e3) * e3_t * Cp[:, None] + f3 * Cp_t[:, None]).view(P + 1, 3, Fh)
        s, s_t = quad(X), (dquad(X) * X_t).sum(1)
        sp1 = (s + 1)[:, None, :]
        Xh = X / sp1
        Xh_t = X_t / sp1 - X * s_t[:, None, :] / sp1 ** 2
        Pn, Pn_t = tlin(Xh, V[0:3]), tlin(Xh_t, V[0:3])
        Mi = A.csr_gather_sum(g, Pn, w)
        Mi_t = A.csr_gather_sum(g, Pn_t, w) + A.csr_gather_sum(g, Pn, w_t)
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
                          e3_t=e3_t, he1=he1, he1_t=he1_t, he2=he2, he2_t=he2_t, w=w, w_t=w_t, V=V, M1=M1, M2=M2, M3=M3))
        inter.update({f"l{l}.w_t": w_t, f"l{l}.Xh_t": Xh_t, f"l{l}.Pn_t": Pn_t, f"l{l}.Mi_t": Mi_t, f"l{l}.Ch_t": Ch_t, f"l{l}.D_t": D_t,
                      f"l{l}.Xn_t": Xn_t})
        X, X_t = Xn, Xn_t

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
    ea = (Fn.silu(ao) @ O2.t() + bO2) * std
    ea_t = ((silu_d1(ao) * ao_t) @ O2.t()) * std
    E = torch.zeros(nmol, 1, dtype=dt).index_add(0, batch, ea) + sd.get("mean", torch.zeros((), dtype=dt)).to(dt)
    s_val = ea_t.sum()
    inter.update(feat_t=feat_t, lnr_t=lnr_t, al_t=al_t, ao_t=ao_t, ea_t=ea_t)

    # ================= reverse pass (seed 1 on every atom's energy) and its tangent
    g_ao, g_ao_t = (std * O2).expand(n, -1) * silu_d1(ao), (std * O2).expand(n, -1) * silu_d2(ao) * ao_t
    ent["O2"] = std * (silu_d1(ao) * ao_t).sum(0, keepdim=True)
    ent["bO2"] = torch.zeros(1, dtype=dt)
    if ge is not None:  # seed - ge[molecule] of the tangent adjoint at every atom's energy
        wa = ge.to(dt).reshape(-1)[batch][:, None]
        g_ao_t = g_ao_t - wa * g_ao
        ent["O2"] = ent["O2"] - std * (wa * Fn.silu(ao)).sum(0, keepdim=True)
        ent["bO2"] = ent["bO2"] - std * wa.sum()
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
        g_Pn = compose_T(g_Y) + A.csr_gather_sum(g, g_Mi, c["w"])
        g_Pn_t = compose_T(g_Y_t) + A.csr_gather_sum(g, g_Mi_t, c["w"]) + A.csr_gather_sum(g, g_Mi, c["w_t"])
        # per-pair weight gradient [P + 1, 3F]; the self pair (row P) collects every atom's self edge
        Pn, Pn_t = c["Pn"], c["Pn_t"]

        def pair_rows(gm, pn):
            return torch.cat([tsum(gm[pi] * pn[pj] + gm[pj] * pn[pi]), tsum(gm * pn).sum(0, keepdim=True)], 0).reshape(P + 1, 3 * Fh)

        g_w, g_w_t = pair_rows(g_Mi, Pn), pair_rows(g_Mi_t, Pn) + pair_rows(g_Mi, Pn_t)
        e1, e2, e3, e1_t, e2_t, e3_t = c["e1"], c["e2"], c["e3"], c["e1_t"], c["e2_t"], c["e3_t"]
        f3 = Fn.silu(e3)
        g_C += (g_w * f3).sum(-1)
        g_C_t += (g_w_t * f3 + g_w * silu_d1(e3) * e3_t).sum(-1)
        C1, C1_t = Cp[:, None], Cp_t[:, None]
        g_e3 = g_w * C1 * silu_d1(e3)
        g_e3_t = g_w_t * C1 * silu_d1(e3) + g_w * C1_t * silu_d1(e3) + g_w * C1 * silu_d2(e3) * e3_t
        ent[f"l{l}.M2"], ent[f"l{l}.b2"] = g_e3_t.t() @ c["he2"] + g_e3.t() @ c["he2_t"], g_e3_t.sum(0)
        g_he2, g_he2_t = g_e3 @ c["M3"], g_e3_t @ c["M3"]
        g_e2, g_e2_t = g_he2 * silu_d1(e2), g_he2_t * silu_d1(e2) + g_he2 * silu_d2(e2) * e2_t
        ent[f"l{l}.M1"], ent[f"l{l}.b1"] = g_e2_t.t() @ c["he1"] + g_e2.t() @ c["he1_t"], g_e2_t.sum(0)
        g_he1, g_he1_t = g_e2 @ c["M2"], g_e2_t @ c["M2"]
        g_e1, g_e1_t = g_he1 * silu_d1(e1), g_he1_t * silu_d1(e1) + g_he1 * silu_d2(e1) * e1_t
        ent[f"l{l}.M0"], ent[f"l{l}.b0"] = g_e1_t.t() @ phi + g_e1.t() @ phi_t, g_e1_t.sum(0)
        g_phi += g_e1 @ c["M1"]
        g_phi_t += g_e1_t @ c["M1"]
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
    g_pos = torch.zeros(n, 3, dtype=dt).index_add(0, pi, g_delta).index_add(0, pj, -g_delta)
    Hv = torch.zeros(n, 3, dtype=dt).index_add(0, pi, g_delta_t).index_add(0, pj, -g_delta_t)
    inter.update(g_d_t=g_d_t, g_rh_t=g_rh_t)
    return dict(E=E, F=-g_pos, s=s_val, ent=ent, Hv=Hv, inter=inter)


def state_dict_grads(ent, sd, hp):
    """engine entries -> gradients by state-dict key (the chain rule of torchmdnet_amd/models/model.py _tensornet_grads)."""
    R, O = "representation_model.", "output_model.output_network.layers."
    T = R + "tensor_embedding."
    Fh, L = hp["hidden_channels"], hp["num_layers"]
    out = {T + f"linears_tensor.{k}.weight": ent[f"Ue{k}"] for k in range(3)}
    out.update({T + "linears_scalar.0.weight": ent["L1"], T + "linears_scalar.0.bias": ent["bL1"], T + "linears_scalar.1.weight": ent["L2"],
                T + "linears_scalar.1.bias": ent["bL2"], T + "init_norm.weight": ent["ln0_w"], T + "init_norm.bias": ent["ln0_b"],
                R + "out_norm.weight": ent["lnr_w"], R + "out_norm.bias": ent["lnr_b"], R + "linear.weight": ent["Lin"],
                R + "linear.bias": ent["bLin"], O + "0.weight": ent["O1"], O + "0.bias": ent["bO1"], O + "2.weight": ent["O2"],
                O + "2.bias": ent["bO2"]})
    for l in range(L):
        for k in range(3):
            out[R + f"layers.{l}.linears_scalar.{k}.weight"] = ent[f"l{l}.M{k}"]
            out[R + f"layers.{l}.linears_scalar.{k}.bias"] = ent[f"l{l}.b{k}"]
            out[R + f"layers.{l}.linears_tensor.{k}.weight"] = ent[f"l{l}.Va{k}"]
            out[R + f"layers.{l}.linears_tensor.{3 + k}.weight"] = ent[f"l{l}.Vb{k}"]
    Wdp, bdp = ent["Wdp"].view(3, Fh, -1), ent["bdp"].view(3, Fh)
    for k in range(3):
        out[T + f"distance_proj{k + 1}.weight"], out[T + f"distance_proj{k + 1}.bias"] = Wdp[k], bdp[k]
    dU, dV = ent["Utab"], ent["Vtab"]
    emb, w2 = sd[T + "emb.weight"], sd[T + "emb2.weight"]
    out[T + "emb.weight"] = dU @ w2[:, :Fh] + dV @ w2[:, Fh:]
    out[T + "emb2.weight"] = torch.cat([dU.t() @ emb, dV.t() @ emb], dim=1)
    out[T + "emb2.bias"] = dU.sum(0)
    return out
