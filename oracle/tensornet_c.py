"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/tensornet_c.c (plain-C restatement with the
hand-derived reverse pass).  Build with ``make -C oracle``; used by tests/ and bench.py's cpu_baseline."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
MAXL = 16


def _params_struct(real):
    P = C.POINTER(real)
    return type("TnParams", (C.Structure,), {"_fields_": [
        ("F", C.c_int32), ("L", C.c_int32), ("K", C.c_int32), ("max_z", C.c_int32), ("H", C.c_int32), ("o3", C.c_int32),
        ("has_atomref", C.c_int32),
        ("lo", real), ("up", real), ("mean", real), ("std", real),
        ("means", P), ("betas", P),
        ("dp_w", P * 3), ("dp_b", P * 3),
        ("emb", P), ("emb2_w", P), ("emb2_b", P),
        ("Ue", P * 3),
        ("L1_w", P), ("L1_b", P), ("L2_w", P), ("L2_b", P),
        ("ln0_w", P), ("ln0_b", P),
        ("M_w", (P * 3) * MAXL), ("M_b", (P * 3) * MAXL),
        ("V", (P * 6) * MAXL),
        ("lnr_w", P), ("lnr_b", P), ("Lin_w", P), ("Lin_b", P),
        ("O1_w", P), ("O1_b", P), ("O2_w", P), ("O2_b", P),
        ("atomref", P),
    ]})


_libs = {}


def lib(dtype):
    name = "f64" if dtype == torch.float64 else "f32"
    if name not in _libs:
        path = os.path.join(HERE, "_build", f"libtn_oracle_{name}.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", HERE])
        L = C.CDLL(path)
        real = C.c_double if name == "f64" else C.c_float
        S = _params_struct(real)
        assert L.tn_oracle_sizeof_real() == C.sizeof(real) and L.tn_oracle_sizeof_params() == C.sizeof(S)
        _libs[name] = (L, real, S)
    return _libs[name]


def energy_forces(sd, hp, z, pos, batch, box=None, q=None, atomref=None, want_forces=True):
    """Same contract as oracle/tensornet_torch.energy_and_forces; dtype follows ``pos`` (fp32 or fp64)."""
    dt = pos.dtype
    L, real, S = lib(dt)
    npdt = np.float64 if dt == torch.float64 else np.float32
    keep = []

    def arr(t):
        a = np.ascontiguousarray(t.detach().cpu().numpy().astype(npdt))
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(real))

    R, T = "representation_model.", "representation_model.tensor_embedding."
    p = S()
    p.F, p.L, p.K, p.max_z = hp["hidden_channels"], hp["num_layers"], hp["num_rbf"], hp["max_z"]
    p.H = sd["output_model.output_network.layers.0.weight"].shape[0]
    p.o3 = 1 if hp.get("equivariance_invariance_group", "O(3)") == "O(3)" else 0
    p.has_atomref = 0 if atomref is None else 1
    p.lo, p.up = float(hp["cutoff_lower"]), float(hp["cutoff_upper"])
    p.mean = float(sd["mean"]) if "mean" in sd else 0.0
    p.std = float(sd["std"]) if "std" in sd else 1.0
    p.means, p.betas = arr(sd[R + "distance_expansion.means"]), arr(sd[R + "distance_expansion.betas"])
    for k in range(3):
        p.dp_w[k], p.dp_b[k] = arr(sd[T + f"distance_proj{k + 1}.weight"]), arr(sd[T + f"distance_proj{k + 1}.bias"])
        p.Ue[k] = arr(sd[T + f"linears_tensor.{k}.weight"])
    p.emb, p.emb2_w, p.emb2_b = arr(sd[T + "emb.weight"]), arr(sd[T + "emb2.weight"]), arr(sd[T + "emb2.bias"])
    p.L1_w, p.L1_b = arr(sd[T + "linears_scalar.0.weight"]), arr(sd[T + "linears_scalar.0.bias"])
    p.L2_w, p.L2_b = arr(sd[T + "linears_scalar.1.weight"]), arr(sd[T + "linears_scalar.1.bias"])
    p.ln0_w, p.ln0_b = arr(sd[T + "init_norm.weight"]), arr(sd[T + "init_norm.bias"])
    for l in range(p.L):
        for k in range(3):
            p.M_w[l][k] = arr(sd[R + f"layers.{l}.linears_scalar.{k}.weight"])
            p.M_b[l][k] = arr(sd[R + f"layers.{l}.linears_scalar.{k}.bias"])
        for k in range(6):
            p.V[l][k] = arr(sd[R + f"layers.{l}.linears_tensor.{k}.weight"])
    p.lnr_w, p.lnr_b = arr(sd[R + "out_norm.weight"]), arr(sd[R + "out_norm.bias"])
    p.Lin_w, p.Lin_b = arr(sd[R + "linear.weight"]), arr(sd[R + "linear.bias"])
    O = "output_model.output_network.layers."
    p.O1_w, p.O1_b, p.O2_w, p.O2_b = arr(sd[O + "0.weight"]), arr(sd[O + "0.bias"]), arr(sd[O + "2.weight"]), arr(sd[O + "2.bias"])
    if atomref is not None:
        p.atomref = arr(atomref.reshape(-1))
    n = int(z.shape[0])
    nmol = int(batch.max()) + 1 if n else 0
    zz = np.ascontiguousarray(z.cpu().numpy().astype(np.int64))
    bb = np.ascontiguousarray(batch.cpu().numpy().astype(np.int64))
    pp = np.ascontiguousarray(pos.detach().cpu().numpy().astype(npdt))
    E = np.zeros(nmol, dtype=npdt)
    Fo = np.zeros((n, 3), dtype=npdt)
    box_mode, boxp = 0, None
    if box is not None:
        bx = np.ascontiguousarray(box.detach().cpu().numpy().astype(npdt))
        box_mode = 1 if bx.ndim == 2 else 2
        boxp = bx.ctypes.data_as(C.POINTER(real))
    qp = None
    if q is not None:
        qq = np.ascontiguousarray(q.detach().cpu().numpy().astype(npdt))
        qp = qq.ctypes.data_as(C.POINTER(real))
    rc = L.tn_oracle_energy_forces(C.byref(p), n, nmol, zz.ctypes.data_as(C.c_void_p), pp.ctypes.data_as(C.POINTER(real)),
                                   bb.ctypes.data_as(C.c_void_p), boxp, box_mode, qp, E.ctypes.data_as(C.POINTER(real)),
                                   Fo.ctypes.data_as(C.POINTER(real)) if want_forces else None)
    assert rc >= 0
    return torch.from_numpy(E).view(-1, 1), (torch.from_numpy(Fo) if want_forces else None)
