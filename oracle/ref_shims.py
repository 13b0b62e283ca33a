"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference package from /root/reference.

This module is part of the oracle: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use anything under oracle/.  It is used in the build container (where
/root/reference exists) to (a) validate oracle/tensornet_torch.py and oracle/tensornet_c.c and
(b) generate the golden vectors committed under tests/golden/ (oracle/make_golden.py).  It never
runs on the GPU box (no /root/reference there).

The reference imports two things that are absent from this image:

* ``lightning_utilities.core.rank_zero.rank_zero_warn`` (torchmdnet/models/model.py:16,
  torchmdnet/utils.py:10, torchmdnet/priors/atomref.py:9)  -> stubbed with ``warnings.warn``.
* ``torchmdnet.extensions.ops.get_neighbor_pairs_kernel`` (torchmdnet/models/utils.py:10), which
  pulls in NVIDIA Warp -> replaced by a pure-torch, differentiable brute-force pair search that
  follows the semantics of torchmdnet/extensions/warp_kernels/neighbors_brute.py:98-197 and of the
  executable specification tests/test_neighbors.py:30-71.

With those two shims ``torchmdnet/models/tensornet.py:54-81`` falls into its pure-PyTorch branch
(OPT=False), which is the oracle the HIP path is compared against.
"""
import importlib
import os
import sys
import types
import warnings

import torch

REFERENCE_ROOT = os.environ.get("TMDNET_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "torchmdnet", "models"))


def _apply_pbc(delta, box):
    # triclinic minimum image, z -> y -> x (neighbors_brute.py:116-135, tests/test_neighbors.py:19-27)
    s3 = torch.round(delta[:, 2] / box[:, 2, 2])
    delta = delta - s3[:, None] * box[:, 2]
    s2 = torch.round(delta[:, 1] / box[:, 1, 1])
    delta = delta - s2[:, None] * box[:, 1]
    s1 = torch.round(delta[:, 0] / box[:, 0, 0])
    delta = delta - s1[:, None] * box[:, 0]
    return delta


def shim_get_neighbor_pairs_kernel(
    strategy,
    positions,
    batch,
    box_vectors,
    use_periodic,
    cutoff_lower,
    cutoff_upper,
    max_num_pairs,
    loop,
    include_transpose,
    num_cells,
):
    """Differentiable torch restatement of the brute-force pair list (both strategies give the
    same pair set in the reference: tests/test_neighbors.py:74-148)."""
    n = positions.shape[0]
    dev = positions.device
    # per-molecule blocks so the O((sum N)^2) enumeration is not paid for large batches
    ii, jj = [], []
    b_cpu = batch.detach().cpu()
    order = torch.argsort(b_cpu, stable=True)
    sb = b_cpu[order]
    uniq, counts = torch.unique_consecutive(sb, return_counts=True)
    start = 0
    for c in counts.tolist():
        idx = order[start : start + c]
        tri = torch.tril_indices(c, c, -1)
        ii.append(idx[tri[0]])
        jj.append(idx[tri[1]])
        start += c
    i = torch.cat(ii).to(dev)
    j = torch.cat(jj).to(dev)
    # canonical orientation i > j as in the triangular index of neighbors_brute.py:63-96
    swap = i < j
    i, j = torch.where(swap, j, i), torch.where(swap, i, j)
    delta = positions[i] - positions[j]
    if use_periodic:
        box = box_vectors.to(positions.dtype)
        if box.dim() == 2:
            box = box.unsqueeze(0).expand(int(batch.max()) + 1, 3, 3)
        delta = _apply_pbc(delta, box[batch[i]])
    d2 = (delta * delta).sum(-1)
    keep = (d2 < cutoff_upper * cutoff_upper) & (d2 >= cutoff_lower * cutoff_lower)
    i, j, delta = i[keep], j[keep], delta[keep]
    dist = delta.norm(dim=-1)
    nb = [torch.stack([i, j])]
    dl = [delta]
    ds = [dist]
    if include_transpose:
        nb.append(torch.stack([j, i]))
        dl.append(-delta)
        ds.append(dist)
    if loop:
        ar = torch.arange(n, device=dev)
        nb.append(torch.stack([ar, ar]))
        dl.append(torch.zeros(n, 3, dtype=positions.dtype, device=dev))
        ds.append(torch.zeros(n, dtype=positions.dtype, device=dev))
    neighbors = torch.cat(nb, dim=1)
    deltas = torch.cat(dl)
    distances = torch.cat(ds)
    num = neighbors.shape[1]
    num_pairs = torch.tensor([num], dtype=torch.int32, device=dev)
    if num < max_num_pairs:
        pad = max_num_pairs - num
        neighbors = torch.cat(
            [neighbors, torch.full((2, pad), -1, dtype=neighbors.dtype, device=dev)], dim=1
        )
        deltas = torch.cat([deltas, torch.zeros(pad, 3, dtype=deltas.dtype, device=dev)])
        distances = torch.cat([distances, torch.zeros(pad, dtype=distances.dtype, device=dev)])
    elif num > max_num_pairs:
        # the Warp kernels drop writes beyond max_pairs but keep counting
        # (neighbors_brute.py:149); the host then raises (models/utils.py:297-300)
        neighbors = neighbors[:, :max_num_pairs]
        deltas = deltas[:max_num_pairs]
        distances = distances[:max_num_pairs]
    return neighbors, deltas, distances, num_pairs


_installed = False


def install():
    """Install the shims and put /root/reference on sys.path.  Idempotent."""
    global _installed
    if "purge_foreign_torchmdnet" in globals():
        purge_foreign_torchmdnet()
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    # --- lightning_utilities stub
    if "lightning_utilities" not in sys.modules:
        lu = types.ModuleType("lightning_utilities")
        core = types.ModuleType("lightning_utilities.core")
        rz = types.ModuleType("lightning_utilities.core.rank_zero")
        rz.rank_zero_warn = lambda msg, *a, **k: warnings.warn(str(msg))
        lu.core = core
        core.rank_zero = rz
        sys.modules["lightning_utilities"] = lu
        sys.modules["lightning_utilities.core"] = core
        sys.modules["lightning_utilities.core.rank_zero"] = rz
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # --- torchmdnet.extensions stand-in (the real package imports warp at import time)
    import torchmdnet  # noqa: F401  (the reference's top-level package: cheap, no heavy imports)

    ext = types.ModuleType("torchmdnet.extensions")
    ext.__path__ = []  # mark as package; submodule imports other than .ops raise ImportError
    ops = types.ModuleType("torchmdnet.extensions.ops")
    ops.get_neighbor_pairs_kernel = shim_get_neighbor_pairs_kernel
    ext.ops = ops
    sys.modules["torchmdnet.extensions"] = ext
    sys.modules["torchmdnet.extensions.ops"] = ops
    _installed = True


def _is_reference_module(mod) -> bool:
    f = getattr(mod, "__file__", None) or ""
    return os.path.abspath(f).startswith(os.path.abspath(REFERENCE_ROOT) + os.sep)


def purge_foreign_torchmdnet():
    """Drop every ``torchmdnet*`` entry of sys.modules that is not the reference's own file (e.g. the aliases that
    ``torchmdnet_amd.install_as_torchmdnet()`` registers) and forget the installed state, so that the next install() binds
    the real reference again.  Returns the number of entries removed."""
    global _installed
    top = sys.modules.get("torchmdnet")
    mm = sys.modules.get("torchmdnet.models.model")
    clean = (top is None or _is_reference_module(top)) and (mm is None or _is_reference_module(mm))
    if clean:
        return 0
    names = [k for k in sys.modules if k == "torchmdnet" or k.startswith("torchmdnet.")]
    for k in names:
        del sys.modules[k]
    _installed = False
    return len(names)


def reference_model_module():
    purge_foreign_torchmdnet()
    install()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module("torchmdnet.models.model")


def load_example_args(model_name, remove_prior=False, config_file=None, **kwargs):
    """Restatement of tests/utils.py:11-34 (that file imports torch_geometric, absent here)."""
    import yaml

    if config_file is None:
        name = "TensorNet-QM9.yaml" if model_name == "tensornet" else "ET-QM9.yaml"
        config_file = os.path.join(REFERENCE_ROOT, "examples", name)
    with open(config_file, "r") as f:
        args = yaml.load(f, Loader=yaml.FullLoader)
    args.setdefault("precision", 32)
    args["model"] = model_name
    args["seed"] = 1234
    if remove_prior:
        args["prior_model"] = None
    args.setdefault("box_vecs", None)
    args.setdefault("remove_ref_energy", False)
    for key, val in kwargs.items():
        assert key in args, f"unknown key {key}"
        args[key] = val
    return args


def create_example_batch(n_atoms=6, multiple_batches=True):
    """Restatement of tests/utils.py:37-46 (same RNG consumption order)."""
    zs = torch.tensor([1, 6, 7, 8, 9], dtype=torch.long)
    z = zs[torch.randint(0, len(zs), (n_atoms,))]
    pos = torch.randn(len(z), 3)
    batch = torch.zeros(len(z), dtype=torch.long)
    if multiple_batches:
        batch[len(batch) // 2 :] = 1
    return z, pos, batch


def seed_everything(seed):
    """pl.seed_everything equivalent for the generators the fixtures use."""
    import random

    import numpy as np

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
