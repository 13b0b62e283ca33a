"""ctypes binding of libtmdnet_amd.so (include/tmdnet_amd.h).

There is NO CPU fallback: if the shared library is missing or a symbol is absent the import of the
compute path fails loudly.  The library itself needs a gfx950 device only when a compute entry
point is called; dlopen + symbol lookup work on a CPU-only host (used by the "not gpu" tests).
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libtmdnet_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "tmdnet_amd.h")

OK, ERR_INVALID, ERR_HIP, ERR_OVERFLOW, ERR_WORKSPACE, ERR_STATE = 0, 1, 2, 3, 4, 5


class HParams(C.Structure):
    _fields_ = [
        ("hidden_channels", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_rbf", C.c_int32),
        ("max_z", C.c_int32),
        ("max_num_neighbors", C.c_int32),
        ("group_o3", C.c_int32),
        ("head_hidden", C.c_int32),
        ("has_atomref", C.c_int32),
        ("cutoff_lower", C.c_float),
        ("cutoff_upper", C.c_float),
    ]


class EtHParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden_channels", "num_layers", "num_rbf", "max_z", "max_num_neighbors", "num_heads",
                                         "neighbor_embedding", "vector_cutoff", "distance_influence", "has_atomref")] + \
               [("cutoff_lower", C.c_float), ("cutoff_upper", C.c_float)]


class Tn2HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden_channels", "num_layers", "num_rbf", "max_z", "max_num_neighbors", "group_o3",
                                         "head_hidden", "has_atomref", "q_dim")] + \
               [("cutoff_lower", C.c_float), ("cutoff_upper", C.c_float), ("coulomb_cutoff", C.c_float),
                ("coulomb_epsilon_solvent", C.c_float)]


_lib = None


def lib():
    """Load the shared library (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python __graft_entry__.py). "
            "torchmdnet_amd has no CPU or eager-PyTorch fallback."
        )
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
    L.tmdnet_create.argtypes = [C.POINTER(HParams), C.POINTER(vp)]
    L.tmdnet_create_et.argtypes = [C.POINTER(EtHParams), C.POINTER(vp)]
    L.tmdnet_create_tn2.argtypes = [C.POINTER(Tn2HParams), C.POINTER(vp)]
    L.tmdnet_destroy.argtypes = [vp]
    L.tmdnet_last_error.argtypes = [vp]
    L.tmdnet_last_error.restype = C.c_char_p
    L.tmdnet_version.restype = C.c_char_p
    L.tmdnet_set_param.argtypes = [vp, C.c_char_p, vp, i64]
    L.tmdnet_finalize_params.argtypes = [vp]
    L.tmdnet_update_params_device.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(vp)]
    L.tmdnet_num_params.argtypes = [vp]
    L.tmdnet_param_name.argtypes = [vp, C.c_int, C.POINTER(i64)]
    L.tmdnet_param_name.restype = C.c_char_p
    L.tmdnet_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.tmdnet_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    L.tmdnet_graph_workspace_bytes.argtypes = [vp, i64, i64, C.POINTER(sz)]
    L.tmdnet_build_graph.argtypes = [vp, vp, vp, sz, i64, i64, vp, vp, vp, vp, i32, C.POINTER(i64)]
    L.tmdnet_set_cell_grid.argtypes = [vp, i32, i32, i32]
    L.tmdnet_set_atom_weights.argtypes = [vp, vp]
    L.tmdnet_set_halo_exchange.argtypes = [vp, HALO_EXCHANGE_FN, vp]
    L.tmdnet_build_graph_static.argtypes = [vp, vp, vp, sz, i64, i64, vp, vp, vp, vp, i32]
    L.tmdnet_graph_counts.argtypes = [vp, vp, vp, i64, i64, C.POINTER(i64)]
    L.tmdnet_graph_cell_grid.argtypes = [vp, vp, vp, i64, i64, C.POINTER(i64)]
    L.tmdnet_forward_workspace_bytes.argtypes = [vp, i64, i64, i64, i64, i32, C.POINTER(sz)]
    L.tmdnet_energy_forces.argtypes = [vp, vp, vp, vp, sz, i64, i64, i64, vp, vp, vp, i32, vp, vp]
    L.tmdnet_neighbor_workspace_bytes.argtypes = [i64, i64, i64, C.POINTER(sz)]
    L.tmdnet_neighbor_pairs.argtypes = [vp, vp, sz, i64, i64, vp, vp, vp, i32, f32, f32, i64, i32, i32, i32, vp, vp, vp, vp]
    L.tmdnet_neighbor_grad.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, vp]
    f64 = C.c_double
    L.tmdnet_neighbor_pairs_f64.argtypes = [vp, vp, sz, i64, i64, vp, vp, vp, i32, f64, f64, i64, i32, i32, vp, vp, vp, vp]
    L.tmdnet_neighbor_grad_f64.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.tmdnet_profile_begin.argtypes = [vp, C.c_uint32]
    L.tmdnet_profile_end.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]
    L.tmdnet_profile_end_records.argtypes = [vp, vp, i64, C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.c_char_p, C.POINTER(i64)]
    L.tmdnet_profile_category_name.argtypes = [C.c_int]
    L.tmdnet_profile_category_name.restype = C.c_char_p
    L.tmdnet_debug_tensor.argtypes = [vp, vp, C.c_char_p, vp, i64]
    L.tmdnet_debug_gemm_dual.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp, vp, vp]
    L.tmdnet_debug_split_weight.argtypes = [vp, i64, i64, vp]
    L.tmdnet_debug_split_weight.restype = i64
    L.tmdnet_debug_gemm.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]
    L.tmdnet_param_grad_count.argtypes = [vp]
    L.tmdnet_param_grad_entry.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.tmdnet_param_grad_entry.restype = C.c_char_p
    L.tmdnet_train_workspace_bytes.argtypes = [vp, i64, i64, i64, C.POINTER(sz), C.POINTER(sz), C.POINTER(i64)]
    L.tmdnet_energy_param_grads.argtypes = [vp, vp, vp, vp, sz, vp, sz, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.tmdnet_force_param_workspace_bytes.argtypes = [vp, i64, i64, i64, C.POINTER(sz)]
    L.tmdnet_force_param_grads.argtypes = [vp, vp, vp, vp, sz, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.tmdnet_loss_param_grads.argtypes = [vp, vp, vp, vp, sz, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.tmdnet_hvp_debug_tensor.argtypes = [vp, vp, C.c_char_p, vp, i64]
    abi = int(re.search(r"#define\s+TMDNET_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read()).group(1))
    if L.tmdnet_abi_version() != abi:
        raise ImportError(f"{LIB_PATH} has ABI revision {L.tmdnet_abi_version()}, include/tmdnet_amd.h declares {abi}: rebuild")
    for name in declared_symbols():
        fn = getattr(L, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    _lib = L
    return L


# tmdnet_halo_exchange_fn (include/tmdnet_amd.h): (user, stage, rows, n_rows, row_floats, perm, stream) -> 0 on success
HALO_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)


def declared_symbols():
    """Every function declared in include/tmdnet_amd.h."""
    with open(HEADER_PATH) as fh:
        txt = fh.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tmdnet_[a-z0-9_]+)\s*\(", txt)))


def check_symbols():
    L = lib()
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    if missing:
        raise ImportError(f"libtmdnet_amd.so does not export: {missing}")
    return declared_symbols()
