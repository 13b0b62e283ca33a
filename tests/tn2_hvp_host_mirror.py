"""TEST INFRASTRUCTURE ONLY: the schedule of an analytic second-order pass for TensorNet2 + ScalarPlusWeightedCoulomb, launch by
launch, on the CPU - tests/hvp_host_mirror.py (TensorNet) with what the architecture adds: the ChargePredict heads, the edge MLP per
DIRECTED edge (rows = CSR edges), the Coulomb head.  Per-element kernels = the bodies of csrc/tn_hvp_math.h and csrc/tn_tn2_hvp_math.h
(compiled host-only, tests/hvp_host.hip); dense products = torch matmuls.  tests/test_hvp_host.py compares with
oracle/tn2_second_order.py.  (Generated from the TensorNet mirror by hand-checked edits: keep the two in step.)"""
import ctypes as C

import torch
import torch.nn.functional as Fn

from oracle import tensornet_adjoint as A
from oracle import tensornet_second_order as S2
from tests.hvp_host_mirror import call, f32


def force_term_mirror(sd, hp, z, pos, batch, v, box=None, q=None, ge=None):
    """-> TensorNet-style entries + the charge heads' gradients by state-dict key, in fp32, by the planned engine schedule."""
    R = "representation_model."
    T = R + "tensor_embedding."
    O = "output_model.output_network.layers."
    sd = {k: (t.float() if t.is_floating_point() else t) for k, t in sd.items()}
    F, L, K = hp["hidden_channels"], hp["num_layers"], hp["num_rbf"]
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    o3 = int(hp.get("equivariance_invariance_group", "O(3)") == "O(3)")
    g = A.build_graph(pos.float(), batch, lo, up, None if box is None else box.float())
    N, P = g["n"], g["P"]
    P1 = P + 1
    i32 = lambda t: t.to(torch.int32).contiguous()
    rowptr, col, epair, esign = i32(g["rowptr"]), i32(g["cols"]), i32(g["epair"]), g["esign"].float().contiguous()
    pair_i, pair_j = i32(g["pi"]), i32(g["pj"])
    pd = torch.cat([g["d"].float(), torch.zeros(1)]).contiguous()
    prhat = g["rhat"].float().contiguous()
    z = z.to(torch.int64).contiguous()
    v = v.float().contiguous()
    kap = None  # TensorNet2 has no charge factor in the group product / update
    qd = hp["q_dim"]
    QC = (L + 1) * qd
    nmol = int(batch.max()) + 1
    Qmol = None if q is None else q.float().contiguous()
    batch = batch.to(torch.int64).contiguous()
    counts = torch.bincount(batch, minlength=nmol)
    mend = torch.cumsum(counts, 0)
    mstart, mend = i32(mend - counts), i32(mend)
    NE = int(col.numel())
    erow = torch.repeat_interleave(torch.arange(N), (rowptr[1:] - rowptr[:-1]).long())
    key = {(int(r_), int(c_)): e_ for e_, (r_, c_) in enumerate(zip(erow.tolist(), col.tolist()))}
    erev = i32(torch.tensor([key[(int(c_), int(r_))] for r_, c_ in zip(erow.tolist(), col.tolist())]))  # launch_edge_reverse
    pair_edge = i32(torch.tensor([key[(int(a_), int(b_))] for a_, b_ in zip(pair_i.tolist(), pair_j.tolist())]))
    charges, charges_t = torch.zeros(N, QC), torch.zeros(N, QC)
    cps = []

    def charge_head(pre, Xl, Xl_t, k):
        """ChargePredict + equilibration on X -> column block k of `charges`; what the reverse pass needs is kept in cps[k]"""
        cf, cf_t = f32(N, 3 * F), f32(N, 3 * F)
        call("h2_cp_feat_dual", N, F, Xl, Xl_t, cf, cf_t)
        lw, lb = sd[pre + "q_norm.weight"].contiguous(), sd[pre + "q_norm.bias"].contiguous()
        ln_, xh_, rs_, ln_t_, xh_t_, rs_t_ = f32(N, 3 * F), f32(N, 3 * F), f32(N), f32(N, 3 * F), f32(N, 3 * F), f32(N)
        call("hh_ln_dual", N, 3 * F, cf, cf_t, lw, lb, ln_, xh_, rs_, ln_t_, xh_t_, rs_t_)
        ks = [0]
        while (pre + f"q_mlp.layers.{ks[-1] + 2}.weight") in sd:
            ks.append(ks[-1] + 2)
        hs, hs_t, pres, pres_t = [ln_], [ln_t_], [], []
        for kk in ks[:-1]:
            Wk, bk = sd[pre + f"q_mlp.layers.{kk}.weight"], sd[pre + f"q_mlp.layers.{kk}.bias"]
            a_ = gemm(hs[-1], Wk, bk)
            a_t_ = gemm(hs_t[-1], Wk)
            h_t_ = f32(*a_.shape)
            call("hh_silu_tangent", C.c_int64(a_.numel()), a_, a_t_, h_t_)
            pres.append(a_); pres_t.append(a_t_); hs.append(Fn.silu(a_)); hs_t.append(h_t_)
        Wk, bk = sd[pre + f"q_mlp.layers.{ks[-1]}.weight"], sd[pre + f"q_mlp.layers.{ks[-1]}.bias"]
        out_, out_t_ = gemm(hs[-1], Wk, bk), gemm(hs_t[-1], Wk)
        sums = f32(nmol, 4, qd)
        call("h2_cp_mol_sums", nmol, qd, mstart, mend, out_, out_t_, sums)
        call("h2_cp_qeq_dual", N, qd, batch, Qmol, out_, out_t_, sums, charges, charges_t, QC, k * qd)
        cps.append(dict(pre=pre, X=Xl, X_t=Xl_t, xh=xh_, xh_t=xh_t_, rstd=rs_, rstd_t=rs_t_, lw=lw, ks=ks, hs=hs, hs_t=hs_t, pres=pres,
                        pres_t=pres_t, out=out_, out_t=out_t_, sums=sums))

    def charge_head_bwd(k, Gl, Gl_t):
        """adjoint of charge head k (g_ch: column block k of g_charges) added into the residual stream's adjoint (Gl, Gl_t)"""
        S = cps[k]
        pre = S["pre"]
        bs, g_out, g_out_t = f32(nmol, 4, qd), f32(N, 2 * qd), f32(N, 2 * qd)
        call("h2_cp_qeq_bwd_dual", N, nmol, qd, mstart, mend, batch, Qmol, S["out"], S["out_t"], S["sums"], g_charges, g_charges_t, QC, k * qd,
             bs, g_out, g_out_t)
        ks = S["ks"]

        def lin_grad(key_, gy, gy_t, x_, x_t_):
            ext[key_ + ".weight"] = ext.get(key_ + ".weight", 0) + tn_gemm(gy_t, x_) + tn_gemm(gy, x_t_)
            ext[key_ + ".bias"] = ext.get(key_ + ".bias", 0) + gy_t.sum(0)

        lin_grad(pre + f"q_mlp.layers.{ks[-1]}", g_out, g_out_t, S["hs"][-1], S["hs_t"][-1])
        Wk = sd[pre + f"q_mlp.layers.{ks[-1]}.weight"]
        g_h, g_h_t = gemmT(g_out, Wk), gemmT(g_out_t, Wk)
        for i_ in reversed(range(len(ks) - 1)):
            g_a, g_a_t = f32(*g_h.shape), f32(*g_h.shape)
            call("hh_dsilu_dual", C.c_int64(g_h.numel()), g_h, g_h_t, S["pres"][i_], S["pres_t"][i_], g_a, g_a_t)
            lin_grad(pre + f"q_mlp.layers.{ks[i_]}", g_a, g_a_t, S["hs"][i_], S["hs_t"][i_])
            Wk = sd[pre + f"q_mlp.layers.{ks[i_]}.weight"]
            g_h, g_h_t = gemmT(g_a, Wk), gemmT(g_a_t, Wk)
        ext[pre + "q_norm.weight"] = ext.get(pre + "q_norm.weight", 0) + (g_h_t * S["xh"]).sum(0) + (g_h * S["xh_t"]).sum(0)
        ext[pre + "q_norm.bias"] = ext.get(pre + "q_norm.bias", 0) + g_h_t.sum(0)
        g_cf, g_cf_t = f32(N, 3 * F), f32(N, 3 * F)
        call("hh_lnbwd_dual", N, 3 * F, g_h, g_h_t, S["xh"], S["xh_t"], S["rstd"], S["rstd_t"], S["lw"], g_cf, g_cf_t)
        call("h2_cp_feat_bwd_dual", N, F, S["X"], S["X_t"], g_cf, g_cf_t, Gl, Gl_t)

    ext = {}
    H = sd[O + "0.weight"].shape[0]
    std = float(sd["std"]) if "std" in sd else 1.0

    # dense products of the engine: gemm(A, W, bias) = A W^T + bias ; tensor linears ; weight-gradient products
    gemm = lambda a, W, b=None: (a @ W.t() + (0 if b is None else b)).contiguous()
    gemmT = lambda a, W: (a @ W).contiguous()  # the engine passes the stored transpose W^T to gemm()
    tlin = lambda u, Ws, tr=False: A.tensor_linear(u, Ws, tr).contiguous()
    tn_gemm = lambda a, b: a.t() @ b  # out[n][k] = sum_r a[r][n] b[r][k]
    # the engine's plain CSR sweep (launch_message_adjoint): out[i, c] = sum_{e in row i} w[pair(e), type(c)] src[col(e), c]
    sweep = lambda w_, src: A.csr_gather_sum(g, src, w_.view(P1, 3, F)).contiguous()
    C0, NC = (0, 1, 4), (1, 3, 5)

    def tlin_grad(g_out, inp):
        return [sum(g_out[:, c].t() @ inp[:, c] for c in range(C0[k], C0[k] + NC[k])) for k in range(3)]

    ent = {}
    # ---- radial functions per pair (launch_radial) and the tangent of the geometry
    phi, dphi = A.rbf_and_grad(pd, sd[R + "distance_expansion.means"], sd[R + "distance_expansion.betas"], lo, up)
    Cc, dC = A.cutoff_and_grad(pd, lo, up)
    phi, dphi, Cc, dC = phi.contiguous(), dphi.contiguous(), Cc.contiguous(), dC.contiguous()
    d_t, rhat_t, phi_t, C_t = f32(P1), f32(P1, 3), f32(P1, K), f32(P1)
    call("hh_pair_tangent", P, K, pair_i, pair_j, prhat, pd, v, dphi, dC, d_t, rhat_t, phi_t, C_t)
    # position gradient H v: second derivatives of the radial functions, per-pair accumulators of the distance gradient
    means, betas = sd[R + "distance_expansion.means"].contiguous(), sd[R + "distance_expansion.betas"].contiguous()
    d2phi, d2C = f32(P1, K), f32(P1)
    call("hh_radial2", P, K, pd, means, betas, lo, up, d2phi, d2C)
    gC, gC_t, gphid, gphid_t = torch.zeros(P1), torch.zeros(P1), torch.zeros(P1), torch.zeros(P1)

    # ---- embedding, forward
    Wdp = torch.cat([sd[T + f"distance_proj{k}.weight"] for k in (1, 2, 3)], 0)
    bdp = torch.cat([sd[T + f"distance_proj{k}.bias"] for k in (1, 2, 3)], 0)
    W2 = sd[T + "emb2.weight"]
    Utab = (sd[T + "emb.weight"] @ W2[:, :F].t() + sd[T + "emb2.bias"]).contiguous()
    Vtab = (sd[T + "emb.weight"] @ W2[:, F:].t()).contiguous()
    Q, Q_t = gemm(phi, Wdp, bdp), gemm(phi_t, Wdp)
    u0, u0_t, s0n, s0n_t = f32(N, 9, F), f32(N, 9, F), f32(N, F), f32(N, F)
    call("hh_embed_scatter_dual", N, F, P, rowptr, col, epair, esign, z, Utab, Vtab, Q, Q_t, Cc, C_t, prhat, rhat_t, u0, u0_t, s0n, s0n_t)
    ln0_w, ln0_b = sd[T + "init_norm.weight"].contiguous(), sd[T + "init_norm.bias"].contiguous()
    ln0, xh0, rstd0, ln0_t, xh0_t, rstd0_t = f32(N, F), f32(N, F), f32(N), f32(N, F), f32(N, F), f32(N)
    call("hh_ln_dual", N, F, s0n, s0n_t, ln0_w, ln0_b, ln0, xh0, rstd0, ln0_t, xh0_t, rstd0_t)
    L1, bL1 = sd[T + "linears_scalar.0.weight"], sd[T + "linears_scalar.0.bias"]
    L2, bL2 = sd[T + "linears_scalar.1.weight"], sd[T + "linears_scalar.1.bias"]
    a1 = gemm(ln0, L1, bL1)
    h1 = Fn.silu(a1)
    a1_t = gemm(ln0_t, L1)
    h1_t = f32(N, 2 * F)
    call("hh_silu_tangent", C.c_int64(a1.numel()), a1, a1_t, h1_t)
    a2 = gemm(h1, L2, bL2)
    gates = Fn.silu(a2)
    a2_t = gemm(h1_t, L2)
    gates_t = f32(N, 3 * F)
    call("hh_silu_tangent", C.c_int64(a2.numel()), a2, a2_t, gates_t)
    Ue = [sd[T + f"linears_tensor.{k}.weight"] for k in range(3)]
    UX, UX_t = tlin(u0, Ue), tlin(u0_t, Ue)
    X, X_t = [f32(N, 9, F)], [f32(N, 9, F)]
    call("hh_gate_mul_dual", N, F, UX, UX_t, gates, gates_t, X[0], X_t[0])
    charge_head(R + "charge_predict_0.", X[0], X_t[0], 0)

    # ---- interaction layers, forward
    lay = []
    for l in range(L):
        Lp = R + f"layers.{l}."
        M = [sd[Lp + f"linears_scalar.{k}.weight"] for k in range(3)]
        bM = [sd[Lp + f"linears_scalar.{k}.bias"] for k in range(3)]
        V = [sd[Lp + f"linears_tensor.{k}.weight"] for k in range(6)]
        c = dict(M=M, V=V)
        M1a, M1b, M1c = M[0][:, :K].contiguous(), M[0][:, K:K + qd].contiguous(), M[0][:, K + qd:].contiguous()  # the engine's three blocks
        chl, chl_t = charges[:, l * qd:(l + 1) * qd].contiguous(), charges_t[:, l * qd:(l + 1) * qd].contiguous()
        c.update(M1a=M1a, M1b=M1b, M1c=M1c, chl=chl, chl_t=chl_t)
        Ap, Ap_t = gemm(phi, M1a, bM[0]), gemm(phi_t, M1a)          # [P + 1, F]
        Bt, Bt_t, Cs, Cs_t = gemm(chl, M1b), gemm(chl_t, M1b), gemm(chl, M1c), gemm(chl_t, M1c)  # [N, F]
        c["pre1"], c["e1_t"], c["he1"], c["he1_t"] = f32(NE, F), f32(NE, F), f32(NE, F), f32(NE, F)
        call("h2_edge_pre1_dual", NE, N, F, rowptr, col, epair, Ap, Ap_t, Bt, Bt_t, Cs, Cs_t, c["pre1"], c["e1_t"], c["he1"], c["he1_t"])
        c["pre2"] = gemm(c["he1"], M[1], bM[1])
        c["he2"] = Fn.silu(c["pre2"])
        c["e2_t"] = gemm(c["he1_t"], M[1])
        c["he2_t"] = f32(NE, 2 * F)
        call("hh_silu_tangent", C.c_int64(NE * 2 * F), c["pre2"], c["e2_t"], c["he2_t"])
        c["pre3"] = gemm(c["he2"], M[2], bM[2])
        c["e3_t"] = gemm(c["he2_t"], M[2])
        c["w"], c["w_t"] = f32(NE, 3 * F), f32(NE, 3 * F)
        call("h2_w_dual", NE, 3 * F, epair, c["pre3"], c["e3_t"], Cc, C_t, c["w"], c["w_t"])
        c["Xh"], c["Xh_t"] = f32(N, 9, F), f32(N, 9, F)
        call("hh_norm_dual", N, F, X[l], X_t[l], c["Xh"], c["Xh_t"])
        c["Pn"], c["Pn_t"] = tlin(c["Xh"], V[0:3]), tlin(c["Xh_t"], V[0:3])
        c["Mi"], c["Mi_t"] = f32(N, 9, F), f32(N, 9, F)
        call("h2_edge_sweep2", N, F, rowptr, col, None, c["w"], c["Pn"], None, None, None, c["Mi"])
        call("h2_edge_sweep2", N, F, rowptr, col, None, c["w"], c["Pn_t"], c["w_t"], c["Pn"], None, c["Mi_t"])
        c["Ch"], c["Ch_t"] = f32(N, 9, F), f32(N, 9, F)
        call("hh_group_dual", N, F, c["Pn"], c["Pn_t"], c["Mi"], c["Mi_t"], kap, o3, c["Ch"], c["Ch_t"])
        c["D"], c["D_t"] = tlin(c["Ch"], V[3:6]), tlin(c["Ch_t"], V[3:6])
        X.append(f32(N, 9, F))
        X_t.append(f32(N, 9, F))
        call("hh_update_dual", N, F, c["Xh"], c["Xh_t"], c["D"], c["D_t"], kap, X[l + 1], X_t[l + 1])
        lay.append(c)
        charge_head(R + f"charge_predicts.{l}.", X[l + 1], X_t[l + 1], l + 1)

    # ---- readout and head, forward
    feat, feat_t = f32(N, 3 * F), f32(N, 3 * F)
    call("hh_feat_dual", N, F, X[L], X_t[L], feat, feat_t)
    lnr_w, lnr_b = sd[R + "out_norm.weight"].contiguous(), sd[R + "out_norm.bias"].contiguous()
    lnr, xhr, rstdr, lnr_t, xhr_t, rstdr_t = f32(N, 3 * F), f32(N, 3 * F), f32(N), f32(N, 3 * F), f32(N, 3 * F), f32(N)
    call("hh_ln_dual", N, 3 * F, feat, feat_t, lnr_w, lnr_b, lnr, xhr, rstdr, lnr_t, xhr_t, rstdr_t)
    Lin, bLin = sd[R + "linear.weight"], sd[R + "linear.bias"]
    O1, bO1, O2 = sd[O + "0.weight"], sd[O + "0.bias"], sd[O + "2.weight"].reshape(-1).contiguous()
    al = gemm(lnr, Lin, bLin)
    x = Fn.silu(al)
    al_t = gemm(lnr_t, Lin)
    x_t = f32(N, F)
    call("hh_silu_tangent", C.c_int64(N * F), al, al_t, x_t)
    ao, ao_t = gemm(x, O1, bO1), gemm(x_t, O1)
    g_ao, g_ao_t, headv = f32(N, H), f32(N, H), f32(N, H)
    ge32 = None if ge is None else ge.float().contiguous()  # one-pass training: energy seed (tn_hvp_math.h head_dual)
    call("hh_head_dual", C.c_int64(N * H), H, ao, ao_t, O2, std, ge32, batch, g_ao, g_ao_t, headv)
    qw = sd["output_model.qweights"].float().contiguous()
    cut = hp.get("coulomb_cutoff")
    from oracle.tn2_torch import COULOMB_FACTOR
    pos32 = pos.float().contiguous()
    boxf = None if box is None else box.float().contiguous()
    e_c, e_c_t, g_q, g_q_t, gpos_c, hv_c = f32(N), f32(N), f32(N, QC), f32(N, QC), f32(N, 3), f32(N, 3)
    call("h2_coulomb_atom_dual", N, QC, mstart, mend, batch, pos32, v, boxf, 0, charges, charges_t, qw, float(qw.sum()),
         float(cut) if cut is not None else -1.0, float(hp.get("coulomb_epsilon_solvent", 78.3)), float(COULOMB_FACTOR), e_c, e_c_t, g_q, g_q_t,
         gpos_c, hv_c)
    s_val = (headv * O2).sum() + std * e_c_t.sum()
    if ge is not None:  # the Coulomb pairs lie inside a molecule: its adjoints take their atom's factor
        call("hh_row_seed", N, QC, ge32, batch, g_q, g_q_t)
        call("hh_row_seed", N, 3, ge32, batch, gpos_c, hv_c)
    g_charges, g_charges_t = (std * g_q).contiguous(), (std * g_q_t).contiguous()  # adjoints of all charge sets; the edge MLPs add theirs

    # ---- reverse pass with tangents
    ent["O2"], ent["bO2"] = headv.sum(0, keepdim=True), torch.zeros(1)
    if ge is not None:
        call("hh_head_bias_seed", N, std, ge32, batch, ent["bO2"])
    ent["O1"], ent["bO1"] = tn_gemm(g_ao_t, x) + tn_gemm(g_ao, x_t), g_ao_t.sum(0)
    g_x, g_x_t = gemmT(g_ao, O1), gemmT(g_ao_t, O1)
    g_al, g_al_t = f32(N, F), f32(N, F)
    call("hh_dsilu_dual", C.c_int64(N * F), g_x, g_x_t, al, al_t, g_al, g_al_t)
    ent["Lin"], ent["bLin"] = tn_gemm(g_al_t, lnr) + tn_gemm(g_al, lnr_t), g_al_t.sum(0)
    g_ln, g_ln_t = gemmT(g_al, Lin), gemmT(g_al_t, Lin)
    ent["lnr_w"], ent["lnr_b"] = (g_ln_t * xhr).sum(0) + (g_ln * xhr_t).sum(0), g_ln_t.sum(0)
    g_feat, g_feat_t = f32(N, 3 * F), f32(N, 3 * F)
    call("hh_lnbwd_dual", N, 3 * F, g_ln, g_ln_t, xhr, xhr_t, rstdr, rstdr_t, lnr_w, g_feat, g_feat_t)
    G, G_t = f32(N, 9, F), f32(N, 9, F)
    call("hh_readout_bwd_dual", N, F, X[L], X_t[L], g_feat, g_feat_t, G, G_t)
    charge_head_bwd(L, G, G_t)
    gCe, gCe_t = torch.zeros(NE), torch.zeros(NE)  # adjoint of the cutoff factor per directed edge, summed over the layers
    for l in reversed(range(L)):
        c = lay[l]
        V, M = c["V"], c["M"]
        g_D, g_D_t = f32(N, 9, F), f32(N, 9, F)
        call("hh_update_bwd_dual", N, F, G, G_t, c["D"], c["D_t"], kap, g_D, g_D_t)
        for k, (a_, b_) in enumerate(zip(tlin_grad(g_D_t, c["Ch"]), tlin_grad(g_D, c["Ch_t"]))):
            ent[f"l{l}.Vb{k}"] = a_ + b_
        g_Ch, g_Ch_t = tlin(g_D, V[3:6], True), tlin(g_D_t, V[3:6], True)
        g_Mi, g_Mi_t, g_PnY, g_PnY_t = f32(N, 9, F), f32(N, 9, F), f32(N, 9, F), f32(N, 9, F)
        call("hh_group_bwd_dual", N, F, g_Ch, g_Ch_t, c["Pn"], c["Pn_t"], c["Mi"], c["Mi_t"], kap, o3, g_Mi, g_Mi_t, g_PnY, g_PnY_t)
        g_Pn, g_Pn_t = f32(N, 9, F), f32(N, 9, F)  # adjoint wrt the sources: the weights of the REVERSE edge (col <- row)
        call("h2_edge_sweep2", N, F, rowptr, col, erev, c["w"], g_Mi, None, None, g_PnY, g_Pn)
        call("h2_edge_sweep2", N, F, rowptr, col, erev, c["w"], g_Mi_t, c["w_t"], g_Mi, g_PnY_t, g_Pn_t)
        g3, g3_t = f32(NE, 3 * F), f32(NE, 3 * F)
        gcp, gcp_t = f32(NE, F), f32(NE, F)
        call("h2_edge_gw_dual", NE, N, F, rowptr, col, epair, g_Mi, g_Mi_t, c["Pn"], c["Pn_t"], c["pre3"], c["e3_t"], Cc, C_t, g3, g3_t, gcp,
             gcp_t)
        call("h2_edge_rowdot", NE, F, F, epair, gcp, gcp_t, None, None, d_t, 1, gCe, gCe_t)  # adjoint of C per directed edge
        ent[f"l{l}.M2"], ent[f"l{l}.b2"] = tn_gemm(g3_t, c["he2"]) + tn_gemm(g3, c["he2_t"]), g3_t.sum(0)
        gh2, gh2_t = gemmT(g3, M[2]), gemmT(g3_t, M[2])
        g2, g2_t = f32(NE, 2 * F), f32(NE, 2 * F)
        call("hh_dsilu_dual", C.c_int64(NE * 2 * F), gh2, gh2_t, c["pre2"], c["e2_t"], g2, g2_t)
        ent[f"l{l}.M1"], ent[f"l{l}.b1"] = tn_gemm(g2_t, c["he1"]) + tn_gemm(g2, c["he1_t"]), g2_t.sum(0)
        gh1, gh1_t = gemmT(g2, M[1]), gemmT(g2_t, M[1])
        g1, g1_t = f32(NE, F), f32(NE, F)
        call("hh_dsilu_dual", C.c_int64(NE * F), gh1, gh1_t, c["pre1"], c["e1_t"], g1, g1_t)
        gB, gB_t, gCs, gCs_t, gself, gself_t = f32(N, F), f32(N, F), f32(N, F), f32(N, F), f32(N, F), f32(N, F)
        call("h2_edge_reduce_dual", N, F, rowptr, col, erev, g1, g1_t, gB, gB_t, gCs, gCs_t, gself, gself_t)
        gAp, gAp_t = f32(P1, F), f32(P1, F)
        call("h2_pair_reduce_dual", P, F, pair_edge, erev, g1, g1_t, gAp, gAp_t)
        gAp[P], gAp_t[P] = gself.sum(0), gself_t.sum(0)  # the self pair's row: column sum over the atoms' self edges
        ent[f"l{l}.M0"] = torch.cat([tn_gemm(gAp_t, phi) + tn_gemm(gAp, phi_t), tn_gemm(gB_t, c["chl"]) + tn_gemm(gB, c["chl_t"]),
                                     tn_gemm(gCs_t, c["chl"]) + tn_gemm(gCs, c["chl_t"])], 1)  # [F, K + 2 q_dim] = (M0 | M0b | M0c)
        ent[f"l{l}.b0"] = gAp_t.sum(0)
        dAp, d2Ap = gemm(dphi, c["M1a"]), gemm(d2phi, c["M1a"])
        call("hh_pair_rowdot", P, F, gAp, gAp_t, dAp, d2Ap, d_t, 1, gphid, gphid_t)  # (g_Ap M1a) . phi' = g_Ap . (phi' M1a^T)
        g_charges[:, l * qd:(l + 1) * qd] += gemmT(gB, c["M1b"]) + gemmT(gCs, c["M1c"])
        g_charges_t[:, l * qd:(l + 1) * qd] += gemmT(gB_t, c["M1b"]) + gemmT(gCs_t, c["M1c"])
        for k, (a_, b_) in enumerate(zip(tlin_grad(g_Pn_t, c["Xh"]), tlin_grad(g_Pn, c["Xh_t"]))):
            ent[f"l{l}.Va{k}"] = a_ + b_
        gXl, gXl_t = tlin(g_Pn, V[0:3], True), tlin(g_Pn_t, V[0:3], True)
        Gn, Gn_t = f32(N, 9, F), f32(N, 9, F)
        call("hh_norm_bwd_dual", N, F, X[l], X_t[l], G, G_t, gXl, gXl_t, Gn, Gn_t)
        G, G_t = Gn, Gn_t
        charge_head_bwd(l, G, G_t)  # its channels went into the Coulomb term and into this layer's edge MLP

    # ---- embedding adjoint
    g_UX, g_UX_t, g_a2, g_a2_t = f32(N, 9, F), f32(N, 9, F), f32(N, 3 * F), f32(N, 3 * F)
    call("hh_gate_bwd_dual", N, F, G, G_t, UX, UX_t, gates, gates_t, a2, a2_t, g_UX, g_UX_t, g_a2, g_a2_t)
    ent["L2"], ent["bL2"] = tn_gemm(g_a2_t, h1) + tn_gemm(g_a2, h1_t), g_a2_t.sum(0)
    g_h1, g_h1_t = gemmT(g_a2, L2), gemmT(g_a2_t, L2)
    g_a1, g_a1_t = f32(N, 2 * F), f32(N, 2 * F)
    call("hh_dsilu_dual", C.c_int64(N * 2 * F), g_h1, g_h1_t, a1, a1_t, g_a1, g_a1_t)
    ent["L1"], ent["bL1"] = tn_gemm(g_a1_t, ln0) + tn_gemm(g_a1, ln0_t), g_a1_t.sum(0)
    g_ln0, g_ln0_t = gemmT(g_a1, L1), gemmT(g_a1_t, L1)
    ent["ln0_w"], ent["ln0_b"] = (g_ln0_t * xh0).sum(0) + (g_ln0 * xh0_t).sum(0), g_ln0_t.sum(0)
    g_s0n, g_s0n_t = f32(N, F), f32(N, F)
    call("hh_lnbwd_dual", N, F, g_ln0, g_ln0_t, xh0, xh0_t, rstd0, rstd0_t, ln0_w, g_s0n, g_s0n_t)
    for k, (a_, b_) in enumerate(zip(tlin_grad(g_UX_t, u0), tlin_grad(g_UX, u0_t))):
        ent[f"Ue{k}"] = a_ + b_
    g_u0l, g_u0l_t = tlin(g_UX, Ue, True), tlin(g_UX_t, Ue, True)
    gA, gA_t = f32(N, 10, F), f32(N, 10, F)
    call("hh_embed_bwd_atom_dual", N, F, g_u0l, g_u0l_t, u0, u0_t, g_s0n, g_s0n_t, gA, gA_t)
    dir_ = P1 * 3 * F
    gq, gq_t = f32(2, P1, 3 * F), f32(2, P1, 3 * F)
    selfq, selfq_t, gZu_t, gZv_t = f32(N, F), f32(N, F), f32(N, F), f32(N, F)
    call("hh_embed_edge_dual", N, F, P, rowptr, col, epair, esign, z, Utab, Vtab, Q, Q_t, Cc, C_t, prhat, rhat_t, gA, gA_t, gq, gq_t,
         C.c_int64(dir_), selfq, selfq_t, gZu_t, gZv_t)
    dQ, d2Q = gemm(dphi, Wdp), gemm(d2phi, Wdp)
    for k in (0, 1):  # both directions of every pair (rows < P of each block)
        call("hh_pair_rowdot", P, 3 * F, gq[k], gq_t[k], dQ, d2Q, d_t, 1, gphid, gphid_t)
    ec, ec_t = torch.zeros(2, P1, 4), torch.zeros(2, P1, 4)
    call("hh_edge_geom_dual", int(col.numel()), N, F, P, rowptr, col, epair, esign, z, Utab, Vtab, Q, Q_t, Cc, C_t, prhat, rhat_t, gA, gA_t,
         ec, ec_t, C.c_int64(P1 * 4))
    # the layers' per-edge adjoint of C (both directions of a pair; self edges carry no geometry) -> per pair
    call("h2_pair_from_edges", P, pair_edge, erev, gCe, gCe_t, gC, gC_t)
    gdel, gdel_t = f32(max(P, 1), 3), f32(max(P, 1), 3)
    call("hh_geom_dual", P, pd, prhat, d_t, rhat_t, dC, d2C, gC, gC_t, gphid, gphid_t, ec, ec_t, C.c_int64(P1 * 4), gdel, gdel_t)
    g_pos, Hv = f32(N, 3), f32(N, 3)
    call("hh_pair_to_atom", N, P, rowptr, epair, esign, gdel, g_pos)
    call("hh_pair_to_atom", N, P, rowptr, epair, esign, gdel_t, Hv)
    g_pos, Hv = g_pos + std * gpos_c, Hv + std * hv_c  # the Coulomb term's own pair geometry
    dW = tn_gemm(gq_t[0, :P], phi[:P]) + tn_gemm(gq_t[1, :P], phi[:P]) + tn_gemm(gq[0, :P], phi_t[:P]) + tn_gemm(gq[1, :P], phi_t[:P])
    dW[:F] += selfq_t.sum(0)[:, None] * phi[P][None, :]
    db = gq_t[0, :P].sum(0) + gq_t[1, :P].sum(0)
    db[:F] += selfq_t.sum(0)
    ent["Wdp"], ent["bdp"] = dW, db
    nz = sd[T + "emb.weight"].shape[0]
    onehot = Fn.one_hot(z, nz).float()
    ent["Utab"], ent["Vtab"] = tn_gemm(onehot, gZu_t), tn_gemm(onehot, gZv_t)
    # the engine's intermediates by the names tmdnet_hvp_debug_tensor knows, in schedule order (tests/test_gpu_hvp.py walks them; the
    # scratch of the reverse sweep holds its last layer, l = 0, and the adjoint of the charges its final sum)
    bufs, order = {}, []

    def put(name, t):
        bufs[name] = t
        order.append(name)

    for nm, t in (("phi", phi), ("phi_t", phi_t), ("C_t", C_t), ("d2phi", d2phi), ("Q", Q), ("u0", u0), ("u0_t", u0_t), ("X0", X[0]), ("X_t0", X_t[0])):
        put(nm, t)

    def put_head(k):
        S = cps[k]
        c_ = f"cp{k}."
        for nm, t in (("ln", S["hs"][0]), ("ln_t", S["hs_t"][0]), ("xh", S["xh"]), ("rstd_t", S["rstd_t"]), ("a1", S["pres"][0]),
                      ("a1_t", S["pres_t"][0]), ("h1", S["hs"][1]), ("h1_t", S["hs_t"][1]), ("a2", S["pres"][1]), ("a2_t", S["pres_t"][1]),
                      ("h2", S["hs"][2]), ("h2_t", S["hs_t"][2]), ("out", S["out"]), ("out_t", S["out_t"])):
            put(c_ + nm, t)

    put_head(0)
    for l, c in enumerate(lay):
        for nm in ("pre1", "e1_t", "he1", "he1_t", "pre2", "e2_t", "he2_t", "pre3", "e3_t", "w", "w_t", "Xh", "Xh_t", "Pn", "Pn_t", "Mi", "Mi_t", "Ch",
                   "Ch_t", "D", "D_t"):
            put(f"l{l}.{nm}", c[nm])
        put(f"X{l + 1}", X[l + 1])
        put(f"X_t{l + 1}", X_t[l + 1])
        put_head(l + 1)
    for nm, t in (("charges", charges), ("charges_t", charges_t), ("feat", feat), ("lnr_t", lnr_t), ("al", al), ("x_t", x_t), ("ao_t", ao_t),
                  ("headv", headv), ("e_c", e_c), ("e_c_t", e_c_t), ("g_q", g_q), ("g_q_t", g_q_t), ("gpos_c", gpos_c), ("hv_c", hv_c),
                  ("g_feat", g_feat), ("g_feat_t", g_feat_t), ("g_Mi", g_Mi), ("g_Mi_t", g_Mi_t), ("g_Pn", g_Pn), ("g_Pn_t", g_Pn_t), ("g3", g3),
                  ("g3_t", g3_t), ("g2", g2), ("g2_t", g2_t), ("g1", g1), ("g1_t", g1_t), ("gB", gB), ("gB_t", gB_t), ("gCs", gCs), ("gCs_t", gCs_t),
                  ("gself", gself), ("gself_t", gself_t), ("gAp", gAp), ("gAp_t", gAp_t), ("g_charges", g_charges), ("g_charges_t", g_charges_t),
                  ("gCe", gCe), ("gCe_t", gCe_t), ("G_emb", G), ("G_emb_t", G_t), ("g_UX", g_UX), ("g_UX_t", g_UX_t), ("g_a2_t", g_a2_t),
                  ("g_a1_t", g_a1_t), ("g_s0n_t", g_s0n_t), ("gA", gA), ("gA_t", gA_t), ("gC", gC), ("gC_t", gC_t), ("gphid", gphid),
                  ("gphid_t", gphid_t), ("gdel", gdel), ("gdel_t", gdel_t)):
        put(nm, t)
    return dict(ent=ent, extra=ext, s=s_val, Hv=Hv, F=-g_pos, bufs=bufs, order=order, P=P)
