"""TEST INFRASTRUCTURE ONLY: the schedule of an analytic second-order pass for the Equivariant Transformer in the ENGINE'S packed
layouts (csrc/tn_et_api.hip: q | k | value thirds in one [5F] row, filters dk | dvx | dv1 | dv2 per PAIR, head blocks concatenated),
launch by launch, on the CPU: per-element kernels = the bodies of csrc/tn_et_hvp_math.h / tn_hvp_math.h (compiled host-only,
tests/hvp_host.hip), dense products = torch matmuls.  tests/test_hvp_host.py compares the result, mapped back to state-dict keys the
way torchmdnet_amd/models/model.py::_et_grads does, with oracle/et_second_order.py."""
import ctypes as C

import torch
import torch.nn.functional as Fn

from oracle import tensornet_adjoint as A
from tests.hvp_host_mirror import call, f32


def thirds(w, F, hd):
    """rows of a value-type matrix from the reference's per-head layout [H][3][hd] to three contiguous blocks [3][F]"""
    c = torch.arange(F)
    src = torch.cat([(c // hd) * 3 * hd + t * hd + c % hd for t in range(3)])
    return w[src].contiguous(), src


def force_term_mirror(sd, hp, z, pos, batch, v, box=None, ge=None):
    R = "representation_model."
    sd = {k: (t.float() if t.is_floating_point() else t) for k, t in sd.items()}
    F, H, L, K = hp["hidden_channels"], hp["num_heads"], hp["num_layers"], hp["num_rbf"]
    hd, F2, U = F // H, F // 2, F + F // 2
    lo, up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    vc = int(bool(hp["vector_cutoff"]))
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
    std = float(sd["std"]) if "std" in sd else 1.0
    W_ = lambda k: sd[k + ".weight"]
    B_ = lambda k: sd[k + ".bias"]
    gemm = lambda a, W, b=None: (a @ W.t() + (0 if b is None else b)).contiguous()
    gemmT = lambda a, W: (a @ W).contiguous()
    ent = {}

    def dense(key, bkey, gy, gy_t, x, x_t):
        gy, gy_t, x, x_t = (t.reshape(-1, t.shape[-1]) for t in (gy, gy_t, x, x_t))
        ent[key] = ent.get(key, 0) + gy_t.t() @ x + gy.t() @ x_t
        if bkey:
            ent[bkey] = ent.get(bkey, 0) + gy_t.sum(0)

    # ---- radial functions, tangent of the geometry (shared with the TensorNet pass)
    means, betas = sd[R + "distance_expansion.means"].contiguous(), sd[R + "distance_expansion.betas"].contiguous()
    phi, dphi = A.rbf_and_grad(pd, means, betas, lo, up)
    Cc, dC = A.cutoff_and_grad(pd, lo, up)
    phi, dphi, Cc, dC = phi.contiguous(), dphi.contiguous(), Cc.contiguous(), dC.contiguous()
    d_t, rhat_t, phi_t, C_t = f32(P1), f32(P1, 3), f32(P1, K), f32(P1)
    call("hh_pair_tangent", P, K, pair_i, pair_j, prhat, pd, v, dphi, dC, d_t, rhat_t, phi_t, C_t)
    d2phi, d2C = f32(P1, K), f32(P1)
    call("hh_radial2", P, K, pd, means, betas, lo, up, d2phi, d2C)
    g_dphi, g_dphi_t = torch.zeros(P1), torch.zeros(P1)  # sum over everything that reaches d through phi: sum_k g_phi phi'

    # ---- embedding
    emb = W_(R + "embedding").contiguous()
    ne = bool(hp["neighbor_embedding"])
    if ne:
        Np = R + "neighbor_embedding."
        Wn, bn, embN, Wc, bc = W_(Np + "distance_proj"), B_(Np + "distance_proj"), W_(Np + "embedding").contiguous(), W_(Np + "combine"), B_(Np + "combine")
        en, en_t = gemm(phi, Wn, bn), gemm(phi_t, Wn)
        WnC, WnC_t = f32(P1, F), f32(P1, F)
        call("he_rowscale_dual", C.c_int64(P1), F, en, en_t, Cc, C_t, WnC, WnC_t)
        xcat, xcat_t = f32(N, 2 * F), f32(N, 2 * F)
        call("he_nbr_embed_dual", N, F, P, rowptr, col, epair, esign, z, emb, embN, WnC, WnC_t, xcat, xcat_t)
        x, x_t = gemm(xcat, Wc, bc), gemm(xcat_t, Wc)
    else:
        x, x_t = f32(N, F), f32(N, F)
        call("he_embed_dual", N, F, z, emb, x, x_t)
    vec, vec_t = torch.zeros(N, 3, F), torch.zeros(N, 3, F)
    X, X_t, V, V_t, lay = [x], [x_t], [vec], [vec_t], []
    for l in range(L):
        Lp = R + f"attention_layers.{l}."
        c = {}
        wv, vsrc = thirds(W_(Lp + "v_proj"), F, hd)
        c["Wqkv"] = torch.cat([W_(Lp + "q_proj"), W_(Lp + "k_proj"), wv], 0).contiguous()
        c["bqkv"] = torch.cat([B_(Lp + "q_proj"), B_(Lp + "k_proj"), B_(Lp + "v_proj")[vsrc]], 0).contiguous()
        c["Wvp"], c["Wo"], c["bo"] = W_(Lp + "vec_proj"), W_(Lp + "o_proj"), B_(Lp + "o_proj")
        has_dk, has_dv = (Lp + "dk_proj.weight") in sd, (Lp + "dv_proj.weight") in sd
        ws, bs = [], []
        if has_dk:
            ws.append(W_(Lp + "dk_proj"))
            bs.append(B_(Lp + "dk_proj"))
        if has_dv:
            ws.append(W_(Lp + "dv_proj")[vsrc])
            bs.append(B_(Lp + "dv_proj")[vsrc])
        Wd = sum(w.shape[0] for w in ws)
        c["Wd"], c["ok"], c["ov"] = Wd, (0 if has_dk else -1), ((F if has_dk else 0) if has_dv else -1)
        ln_w, ln_b = W_(Lp + "layernorm").contiguous(), B_(Lp + "layernorm").contiguous()
        c["ln_w"] = ln_w
        xt, xh, rstd, xt_t, xh_t, rstd_t = f32(N, F), f32(N, F), f32(N), f32(N, F), f32(N, F), f32(N)
        call("hh_ln_dual", N, F, X[l], X_t[l], ln_w, ln_b, xt, xh, rstd, xt_t, xh_t, rstd_t)
        qkv, qkv_t = gemm(xt, c["Wqkv"], c["bqkv"]), gemm(xt_t, c["Wqkv"])
        vp, vp_t = gemm(V[l], c["Wvp"]), gemm(V_t[l], c["Wvp"])  # [N, 3, 3F]
        if Wd > 0:
            c["Wdkv"], c["bdkv"] = torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous()
            ekv, ekv_t = gemm(phi, c["Wdkv"], c["bdkv"]), gemm(phi_t, c["Wdkv"])
            dkv = Fn.silu(ekv)
            dkv_t = f32(P1, Wd)
            call("hh_silu_tangent", C.c_int64(P1 * Wd), ekv, ekv_t, dkv_t)
        else:
            ekv = ekv_t = None
            dkv, dkv_t = torch.zeros(1), torch.zeros(1)  # never read (ok = ov = -1)
        attn = (qkv, qkv_t, V[l], V_t[l], dkv, dkv_t, Cc, C_t, prhat, rhat_t, F, hd, max(Wd, 1), c["ok"], c["ov"], vc, P)
        xagg, xagg_t, vagg, vagg_t = f32(N, F), f32(N, F), f32(N, 3, F), f32(N, 3, F)
        call("he_attn_fwd_dual", N, H, *attn, rowptr, col, epair, esign, xagg, xagg_t, vagg, vagg_t)
        o, o_t = gemm(xagg, c["Wo"], c["bo"]), gemm(xagg_t, c["Wo"])
        xn, xn_t, vecn, vecn_t, vdot, vdot_t = f32(N, F), f32(N, F), f32(N, 3, F), f32(N, 3, F), f32(N, F), f32(N, F)
        call("he_update_dual", N, F, X[l], X_t[l], V[l], V_t[l], vp, vp_t, o, o_t, vagg, vagg_t, xn, xn_t, vecn, vecn_t, vdot, vdot_t)
        c.update(xt=xt, xt_t=xt_t, xh=xh, xh_t=xh_t, rstd=rstd, rstd_t=rstd_t, vp=vp, vp_t=vp_t, ekv=ekv, ekv_t=ekv_t, attn=attn, xagg=xagg,
                 xagg_t=xagg_t, o=o, o_t=o_t, vdot=vdot, vdot_t=vdot_t)
        lay.append(c)
        X.append(xn); X_t.append(xn_t); V.append(vecn); V_t.append(vecn_t)

    # ---- out_norm and the EquivariantScalar head (models/utils.py:583-655, output_modules.py:120-163)
    O0, O1 = "output_model.output_network.0.", "output_model.output_network.1."
    lno_w, lno_b = W_(R + "out_norm").contiguous(), B_(R + "out_norm").contiguous()
    xf, xfh, rstdf, xf_t, xfh_t, rstdf_t = f32(N, F), f32(N, F), f32(N), f32(N, F), f32(N, F), f32(N)
    call("hh_ln_dual", N, F, X[L], X_t[L], lno_w, lno_b, xf, xfh, rstdf, xf_t, xfh_t, rstdf_t)
    W1u = torch.cat([W_(O0 + "vec1_proj"), W_(O0 + "vec2_proj")], 0).contiguous()
    Wm1, bm1, Wm2, bm2 = W_(O0 + "update_net.layers.0"), B_(O0 + "update_net.layers.0"), W_(O0 + "update_net.layers.2"), B_(O0 + "update_net.layers.2")
    W21, Wn1, bn1 = W_(O1 + "vec1_proj"), W_(O1 + "update_net.layers.0"), B_(O1 + "update_net.layers.0")
    Wn2, bn2 = W_(O1 + "update_net.layers.2")[0].contiguous(), B_(O1 + "update_net.layers.2")[0]
    u12, u12_t = gemm(V[L], W1u), gemm(V_t[L], W1u)  # [N, 3, U]
    hcat, hcat_t = f32(N, 2 * F), f32(N, 2 * F)
    call("he_cat_norm_dual", N, F, xf, xf_t, F, u12, u12_t, U, F, 2 * F, hcat, hcat_t)
    pre1, pre1_t = gemm(hcat, Wm1, bm1), gemm(hcat_t, Wm1)
    m1, m1_t = Fn.silu(pre1), f32(N, F)
    call("hh_silu_tangent", C.c_int64(N * F), pre1, pre1_t, m1_t)
    y, y_t = gemm(m1, Wm2, bm2), gemm(m1_t, Wm2)
    u2, u2_t = u12[..., F:], u12_t[..., F:]  # views with row stride U: pass base pointers of the slices
    u2c, u2c_t = u2.contiguous(), u2_t.contiguous()
    hcat2, hcat2_t, vq, vq_t = f32(N, F), f32(N, F), f32(N, 3, F2), f32(N, 3, F2)
    call("he_head_mid_dual", N, F2, y, y_t, u2c, u2c_t, F2, hcat2, hcat2_t, vq, vq_t)
    w1, w1_t = gemm(vq, W21), gemm(vq_t, W21)
    call("he_cat_norm_dual", N, F2, None, None, F2, w1, w1_t, F2, F2, F, hcat2, hcat2_t)
    pre2, pre2_t = gemm(hcat2, Wn1, bn1), gemm(hcat2_t, Wn1)
    g_pre2, g_pre2_t, headv = f32(N, F2), f32(N, F2), f32(N, F2)
    b64 = batch.to(torch.int64).contiguous()
    ge32 = None if ge is None else ge.float().contiguous()  # one-pass training: energy seed (tn_hvp_math.h head_dual)
    call("hh_head_dual", C.c_int64(N * F2), F2, pre2, pre2_t, Wn2, std, ge32, b64, g_pre2, g_pre2_t, headv)
    s_val = (headv * Wn2).sum()

    # ---- reverse with tangents: head
    ent["Wn2"], ent["bn2"] = headv.sum(0), torch.zeros(1)
    if ge is not None:
        call("hh_head_bias_seed", N, std, ge32, b64, ent["bn2"])
    dense("Wn1", "bn1", g_pre2, g_pre2_t, hcat2, hcat2_t)
    g_h2, g_h2_t = gemmT(g_pre2, Wn1), gemmT(g_pre2_t, Wn1)  # (g_xs | g_n2)
    g_w1, g_w1_t = f32(N, 3, F2), f32(N, 3, F2)
    call("he_norm_bwd_dual", N, F2, g_h2[:, F2:].contiguous(), g_h2_t[:, F2:].contiguous(), F2, w1, w1_t, F2, g_w1, g_w1_t, F2)
    dense("W21", None, g_w1, g_w1_t, vq, vq_t)
    g_vq, g_vq_t = gemmT(g_w1, W21), gemmT(g_w1_t, W21)
    g_y, g_y_t, g_u12, g_u12_t = f32(N, F), f32(N, F), f32(N, 3, U), f32(N, 3, U)
    g_u2, g_u2_t = f32(N, 3, F2), f32(N, 3, F2)
    call("he_head_mid_bwd_dual", N, F2, y, y_t, u2c, u2c_t, F2, g_h2, g_h2_t, g_vq, g_vq_t, g_y, g_y_t, g_u2, g_u2_t, F2)
    dense("Wm2", "bm2", g_y, g_y_t, m1, m1_t)
    g_m1h, g_m1h_t = gemmT(g_y, Wm2), gemmT(g_y_t, Wm2)
    g_m1, g_m1_t = f32(N, F), f32(N, F)
    call("hh_dsilu_dual", C.c_int64(N * F), g_m1h, g_m1h_t, pre1, pre1_t, g_m1, g_m1_t)
    dense("Wm1", "bm1", g_m1, g_m1_t, hcat, hcat_t)
    g_h1, g_h1_t = gemmT(g_m1, Wm1), gemmT(g_m1_t, Wm1)  # (g_xf | g_n1)
    g_u1, g_u1_t = f32(N, 3, F), f32(N, 3, F)
    call("he_norm_bwd_dual", N, F, g_h1[:, F:].contiguous(), g_h1_t[:, F:].contiguous(), F, u12[..., :F].contiguous(), u12_t[..., :F].contiguous(), F,
         g_u1, g_u1_t, F)
    g_u12, g_u12_t = torch.cat([g_u1, g_u2], -1).contiguous(), torch.cat([g_u1_t, g_u2_t], -1).contiguous()
    dense("W1u", None, g_u12, g_u12_t, V[L], V_t[L])
    g_vec, g_vec_t = gemmT(g_u12, W1u), gemmT(g_u12_t, W1u)
    g_xf, g_xf_t = g_h1[:, :F].contiguous(), g_h1_t[:, :F].contiguous()
    ent["lno_w"], ent["lno_b"] = (g_xf_t * xfh + g_xf * xfh_t).sum(0), g_xf_t.sum(0)
    g_x, g_x_t = f32(N, F), f32(N, F)
    call("hh_lnbwd_dual", N, F, g_xf, g_xf_t, xfh, xfh_t, rstdf, rstdf_t, lno_w, g_x, g_x_t)

    # ---- reverse with tangents: attention layers
    slot_dir = P1 * H * 4
    slots, slots_t = torch.zeros(L, 2, P1, H, 4), torch.zeros(L, 2, P1, H, 4)
    for l in reversed(range(L)):
        c = lay[l]
        t_ = f"l{l}."
        Wd = c["Wd"]
        g_o, g_o_t, g_vp, g_vp_t = f32(N, 3 * F), f32(N, 3 * F), f32(N, 3, 3 * F), f32(N, 3, 3 * F)
        call("he_update_bwd_dual", N, F, g_x, g_x_t, g_vec, g_vec_t, c["vp"], c["vp_t"], c["o"], c["o_t"], c["vdot"], c["vdot_t"], g_o, g_o_t,
             g_vp, g_vp_t)
        dense(t_ + "Wo", t_ + "bo", g_o, g_o_t, c["xagg"], c["xagg_t"])
        dense(t_ + "Wvp", None, g_vp, g_vp_t, V[l], V_t[l])
        g_xagg, g_xagg_t = gemmT(g_o, c["Wo"]), gemmT(g_o_t, c["Wo"])
        g_vagg, g_vagg_t = g_vec.clone(), g_vec_t.clone()  # snapshot: the sweeps read it while the source terms are added to g_vec
        g_qkv, g_qkv_t, g_vin, g_vin_t = f32(N, 5 * F), f32(N, 5 * F), f32(N, 3, F), f32(N, 3, F)
        Wdx = max(Wd, 1)
        gq, gq_t, selfq, selfq_t = torch.zeros(2, P1, Wdx), torch.zeros(2, P1, Wdx), torch.zeros(N, Wdx), torch.zeros(N, Wdx)
        call("he_attn_bwd_dual", N, H, *c["attn"], rowptr, col, epair, esign, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv, g_qkv_t, g_vin, g_vin_t,
             gq, gq_t, C.c_int64(P1 * Wdx), selfq, selfq_t, slots[l], slots_t[l], C.c_int64(slot_dir))
        if Wd > 0:
            self_g, self_g_t = selfq.sum(0).contiguous(), selfq_t.sum(0).contiguous()
            g_e, g_e_t = f32(P1, Wd), f32(P1, Wd)
            call("he_filter_gpre_dual", P, Wd, gq, gq_t, C.c_int64(P1 * Wd), self_g, self_g_t, c["ekv"], c["ekv_t"], g_e, g_e_t)
            dense(t_ + "Wdkv", t_ + "bdkv", g_e, g_e_t, phi, phi_t)
            de, d2e = gemm(dphi, c["Wdkv"]), gemm(d2phi, c["Wdkv"])
            call("hh_pair_rowdot", P1, Wd, g_e, g_e_t, de, d2e, d_t, 1, g_dphi, g_dphi_t)
        dense(t_ + "Wqkv", t_ + "bqkv", g_qkv, g_qkv_t, c["xt"], c["xt_t"])
        g_vec = g_vec + g_vin + gemmT(g_vp, c["Wvp"])
        g_vec_t = g_vec_t + g_vin_t + gemmT(g_vp_t, c["Wvp"])
        g_xt, g_xt_t = gemmT(g_qkv, c["Wqkv"]), gemmT(g_qkv_t, c["Wqkv"])
        ent[t_ + "ln_w"], ent[t_ + "ln_b"] = (g_xt_t * c["xh"] + g_xt * c["xh_t"]).sum(0), g_xt_t.sum(0)
        g_ln, g_ln_t = f32(N, F), f32(N, F)
        call("hh_lnbwd_dual", N, F, g_xt, g_xt_t, c["xh"], c["xh_t"], c["rstd"], c["rstd_t"], c["ln_w"], g_ln, g_ln_t)
        g_x, g_x_t = g_x + g_ln, g_x_t + g_ln_t

    # ---- embeddings
    g_cutn, g_cutn_t = torch.zeros(P1), torch.zeros(P1)  # neighbour embedding: adjoint of its cutoff factor per pair
    nz = emb.shape[0]
    onehot = Fn.one_hot(z, nz).float()
    if ne:
        dense("Wc", "bc", g_x, g_x_t, xcat, xcat_t)
        g_xcat, g_xcat_t = gemmT(g_x, Wc), gemmT(g_x_t, Wc)
        ent["emb"] = onehot.t() @ g_xcat_t[:, :F]
        gqn, gqn_t, gZ_t = torch.zeros(2, P1, F), torch.zeros(2, P1, F), f32(N, F)
        call("he_nbr_bwd_dual", N, F, P, rowptr, col, epair, esign, z, embN, WnC, WnC_t, g_xcat, g_xcat_t, gqn, gqn_t, C.c_int64(P1 * F), gZ_t)
        ent["embN"] = onehot.t() @ gZ_t
        g_Wn, g_Wn_t, g_en, g_en_t = torch.zeros(P1, F), torch.zeros(P1, F), torch.zeros(P1, F), torch.zeros(P1, F)
        call("he_nbr_pair_dual", P, F, gqn, gqn_t, C.c_int64(P1 * F), Cc, C_t, g_Wn, g_Wn_t, g_en, g_en_t)
        dense("Wn", "bn", g_en[:P], g_en_t[:P], phi[:P], phi_t[:P])
        den, d2en = gemm(dphi, Wn), gemm(d2phi, Wn)
        call("hh_pair_rowdot", P, F, g_en, g_en_t, den, d2en, d_t, 1, g_dphi, g_dphi_t)
        call("he_pair_rowdot2", P, F, g_Wn, g_Wn_t, en, en_t, 1, g_cutn, g_cutn_t)
    else:
        ent["emb"] = onehot.t() @ g_x_t

    # ---- geometry: forces (value half) and H v
    g_cut, g_cut_t, g_rh, g_rh_t = f32(P1), f32(P1), f32(P1, 3), f32(P1, 3)
    call("he_pair_slots_dual", P, H, L, slots, slots_t, C.c_int64(2 * slot_dir), C.c_int64(slot_dir), g_cut, g_cut_t, g_rh, g_rh_t)
    g_cut[:P] += g_cutn[:P]
    g_cut_t[:P] += g_cutn_t[:P]
    gdel, gdel_t = f32(max(P, 1), 3), f32(max(P, 1), 3)
    call("he_geom_dual", P, pd, prhat, d_t, rhat_t, dC, d2C, g_cut, g_cut_t, g_dphi, g_dphi_t, g_rh, g_rh_t, gdel, gdel_t)
    g_pos, Hv = f32(N, 3), f32(N, 3)
    call("hh_pair_to_atom", N, P, rowptr, epair, esign, gdel, g_pos)
    call("hh_pair_to_atom", N, P, rowptr, epair, esign, gdel_t, Hv)
    # the engine's intermediates by the names tmdnet_hvp_debug_tensor knows (csrc/tn_et_api.hip et_hvp_debug_tensor), in schedule order;
    # the reverse sweep's scratch holds its last layer (l = 0)
    bufs, order = {}, []

    def put(name, t):
        bufs[name] = t
        order.append(name)

    for l in range(L + 1):
        put(f"x{l}", X[l]); put(f"x_t{l}", X_t[l]); put(f"vec{l}", V[l]); put(f"vec_t{l}", V_t[l])
        if l < L:
            c = lay[l]
            put(f"l{l}.qkv", c["attn"][0]); put(f"l{l}.qkv_t", c["attn"][1])
            if c["Wd"] > 0:
                put(f"l{l}.dkv", c["attn"][4]); put(f"l{l}.dkv_t", c["attn"][5])
            put(f"l{l}.xagg", c["xagg"]); put(f"l{l}.xagg_t", c["xagg_t"]); put(f"l{l}.o_t", c["o_t"]); put(f"l{l}.vdot_t", c["vdot_t"])
    for nm, t in (("pre2", pre2), ("pre2_t", pre2_t), ("g_pre2_t", g_pre2_t), ("headv", headv), ("g_qkv", g_qkv), ("g_qkv_t", g_qkv_t),
                  ("g_vin", g_vin), ("g_vin_t", g_vin_t), ("selfq", selfq), ("selfq_t", selfq_t), ("slots", slots),
                  ("slots_t", slots_t), ("g_x", g_x), ("g_x_t", g_x_t), ("g_cut", g_cut), ("g_cut_t", g_cut_t), ("g_rh", g_rh), ("g_rh_t", g_rh_t),
                  ("gdel", gdel), ("gdel_t", gdel_t)):
        put(nm, t)
    if not ne:  # with the neighbour embedding the engine reuses the filter-adjoint rows for that embedding's per-pair adjoints afterwards
        put("gq", gq)
        put("gq_t", gq_t)
    # rows the kernels never write (nor read): the self pair's row of every per-pair block
    never = {"gq": (2, P1), "gq_t": (2, P1), "slots": (2 * L, P1), "slots_t": (2 * L, P1), "g_cut": (1, P1), "g_cut_t": (1, P1), "g_rh": (1, P1),
             "g_rh_t": (1, P1)}
    return dict(ent=ent, s=s_val, Hv=Hv, F=-g_pos, bufs=bufs, order=order, never=never, P=P)


def state_dict_grads(ent, sd, hp):
    """packed entries -> gradients by state-dict key (torchmdnet_amd/models/model.py::_et_grads)"""
    R = "representation_model."
    F, H, L = hp["hidden_channels"], hp["num_heads"], hp["num_layers"]
    hd, F2 = F // H, F // 2
    _, src = thirds(torch.zeros(3 * F, 1), F, hd)

    def unthirds(g):
        out = torch.empty_like(g)
        out[src] = g
        return out

    out = {R + "embedding.weight": ent["emb"], R + "out_norm.weight": ent["lno_w"], R + "out_norm.bias": ent["lno_b"]}
    if hp["neighbor_embedding"]:
        Np = R + "neighbor_embedding."
        out.update({Np + "embedding.weight": ent["embN"], Np + "distance_proj.weight": ent["Wn"], Np + "distance_proj.bias": ent["bn"],
                    Np + "combine.weight": ent["Wc"], Np + "combine.bias": ent["bc"]})
    for l in range(L):
        Lp, t = R + f"attention_layers.{l}.", f"l{l}."
        Wq, bq = ent[t + "Wqkv"], ent[t + "bqkv"]
        out.update({Lp + "layernorm.weight": ent[t + "ln_w"], Lp + "layernorm.bias": ent[t + "ln_b"], Lp + "q_proj.weight": Wq[:F],
                    Lp + "q_proj.bias": bq[:F], Lp + "k_proj.weight": Wq[F:2 * F], Lp + "k_proj.bias": bq[F:2 * F],
                    Lp + "v_proj.weight": unthirds(Wq[2 * F:]), Lp + "v_proj.bias": unthirds(bq[2 * F:]), Lp + "vec_proj.weight": ent[t + "Wvp"],
                    Lp + "o_proj.weight": ent[t + "Wo"], Lp + "o_proj.bias": ent[t + "bo"]})
        if (t + "Wdkv") in ent:
            Wd, bd, o = ent[t + "Wdkv"], ent[t + "bdkv"], 0
            if (Lp + "dk_proj.weight") in sd:
                out[Lp + "dk_proj.weight"], out[Lp + "dk_proj.bias"] = Wd[:F], bd[:F]
                o = F
            if (Lp + "dv_proj.weight") in sd:
                out[Lp + "dv_proj.weight"], out[Lp + "dv_proj.bias"] = unthirds(Wd[o:o + 3 * F]), unthirds(bd[o:o + 3 * F])
    O0, O1 = "output_model.output_network.0.", "output_model.output_network.1."
    W1u = ent["W1u"]
    w2 = torch.zeros_like(sd[O1 + "update_net.layers.2.weight"], dtype=torch.float32)
    w2[0] = ent["Wn2"]
    out.update({O0 + "vec1_proj.weight": W1u[:F], O0 + "vec2_proj.weight": W1u[F:], O0 + "update_net.layers.0.weight": ent["Wm1"],
                O0 + "update_net.layers.0.bias": ent["bm1"], O0 + "update_net.layers.2.weight": ent["Wm2"],
                O0 + "update_net.layers.2.bias": ent["bm2"], O1 + "vec1_proj.weight": ent["W21"], O1 + "update_net.layers.0.weight": ent["Wn1"],
                O1 + "update_net.layers.0.bias": ent["bn1"], O1 + "update_net.layers.2.weight": w2,
                O1 + "update_net.layers.2.bias": torch.cat([ent["bn2"].reshape(1),
                                                             torch.zeros(sd[O1 + "update_net.layers.2.bias"].numel() - 1)])})
    return out
