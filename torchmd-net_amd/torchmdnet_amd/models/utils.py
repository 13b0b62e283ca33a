"""Host-side mirrors of the reference's torchmdnet/models/utils.py classes that sit on the hot path.

These modules own parameters/buffers under the reference's state-dict names so that existing
checkpoints load unchanged; the arithmetic itself runs in the HIP library (torchmdnet_amd._C).
Standalone callables (OptimizedDistance) go through the library's fine-grained operator.
"""
import ctypes as C
import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from torchmdnet_amd import _C


def _require_cuda(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"torchmdnet_amd: {what} runs only on an AMD GPU through the HIP extension; got a {t.device} tensor. "
            "There is no CPU / eager fallback by design."
        )


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class CosineCutoff(nn.Module):
    """Parameter-free; evaluated inside the radial kernel (reference models/utils.py:500-528)."""

    def __init__(self, cutoff_lower=0.0, cutoff_upper=5.0):
        super().__init__()
        self.cutoff_lower = cutoff_lower
        self.cutoff_upper = cutoff_upper


class ExpNormalSmearing(nn.Module):
    """Owns ``means`` / ``betas`` (reference models/utils.py:356-407); evaluated in the radial kernel."""

    def __init__(self, cutoff_lower=0.0, cutoff_upper=5.0, num_rbf=50, trainable=True, dtype=torch.float32):
        super().__init__()
        self.cutoff_lower = cutoff_lower
        self.cutoff_upper = cutoff_upper
        self.num_rbf = num_rbf
        self.trainable = trainable
        self.dtype = dtype
        self.alpha = 5.0 / (cutoff_upper - cutoff_lower)
        means, betas = self._initial_params()
        if trainable:
            self.register_parameter("means", nn.Parameter(means))
            self.register_parameter("betas", nn.Parameter(betas))
        else:
            self.register_buffer("means", means)
            self.register_buffer("betas", betas)

    def _initial_params(self):
        # PhysNet defaults, same formulas as the reference (models/utils.py:382-395)
        start = torch.exp(torch.scalar_tensor(-self.cutoff_upper + self.cutoff_lower, dtype=self.dtype))
        means = torch.linspace(start, 1, self.num_rbf, dtype=self.dtype)
        betas = torch.tensor([(2 / self.num_rbf * (1 - start)) ** -2] * self.num_rbf, dtype=self.dtype)
        return means, betas

    def reset_parameters(self):
        means, betas = self._initial_params()
        self.means.data.copy_(means)
        self.betas.data.copy_(betas)


rbf_class_mapping = {"expnorm": ExpNormalSmearing}
act_class_mapping = {"silu": nn.SiLU}
dtype_mapping = {16: torch.float16, 32: torch.float, 64: torch.float64}


class MLP(nn.Module):
    """Parameter container with the reference's layout ``layers.{0,2,..}`` (models/utils.py:531-580)."""

    def __init__(self, in_channels, out_channels, hidden_channels, activation, num_hidden_layers=0, dtype=torch.float32):
        super().__init__()
        self.act = act_class_mapping[activation]()
        self.layers = nn.Sequential()
        self.layers.append(nn.Linear(in_channels, hidden_channels, dtype=dtype))
        self.layers.append(self.act)
        for _ in range(num_hidden_layers):
            self.layers.append(nn.Linear(hidden_channels, hidden_channels, dtype=dtype))
            self.layers.append(self.act)
        self.layers.append(nn.Linear(hidden_channels, out_channels, dtype=dtype))

    def reset_parameters(self):
        for layer in self.layers:
            if isinstance(layer, nn.Linear):
                nn.init.xavier_uniform_(layer.weight)
                layer.bias.data.fill_(0)


class OptimizedDistance(nn.Module):
    """Neighbour list with the reference's interface and conventions (models/utils.py:120-313):
    ``edge_vec = pos[edge_index[0]] - pos[edge_index[1]]`` (+ triclinic minimum image), pairs padded
    with (-1,-1) unless ``resize_to_fit``, RuntimeError when more than ``max_num_pairs`` are found.
    ``strategy="brute"``: molecule-restricted wave-per-atom search (tn_graph_wave.hip); ``strategy="cell"``: the O(N) cell
    list (tn_cell.hip) over all molecules at once - one periodic box, or without a box a fictitious one around the
    bounding box of the positions (the reference uses a fixed 3*cutoff box there, models/utils.py:206-212); one box per
    molecule has no common grid and takes the brute-force kernel.  Same pair set, deterministic order, either way."""

    def __init__(self, cutoff_lower=0.0, cutoff_upper=5.0, max_num_pairs=-32, return_vecs=False, loop=False,
                 strategy="brute", include_transpose=True, resize_to_fit=True, box=None, long_edge_index=True):
        super().__init__()
        if strategy not in ("brute", "cell"):
            raise ValueError(f"unknown strategy {strategy}")
        self.cutoff_upper = cutoff_upper
        self.cutoff_lower = cutoff_lower
        self.max_num_pairs = max_num_pairs
        self.strategy = strategy
        self.loop = loop
        self.return_vecs = return_vecs
        self.include_transpose = include_transpose
        self.resize_to_fit = resize_to_fit
        self.use_periodic = box is not None
        self.long_edge_index = long_edge_index
        if box is None:
            box = torch.zeros((3, 3), device="cpu")
        self.register_buffer("box", box, persistent=True)

    def forward(self, pos: Tensor, batch: Optional[Tensor] = None, box: Optional[Tensor] = None
                ) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        if pos.device.type == "cpu" and torch.cuda.is_available():
            # host tensors (the reference's tests build them on the CPU by default): staged to the current AMD GPU, the pair list is
            # built by the HIP kernels there and comes back as host tensors; the moves are autograd-visible.  Not a CPU path:
            # without a GPU the call fails below.
            dev = torch.device("cuda", torch.cuda.current_device())
            out = self.forward(pos.to(dev), None if batch is None else batch.to(dev), None if box is None else box.to(dev))
            return tuple(None if t is None else t.to(pos.device) for t in out)
        _require_cuda(pos, "OptimizedDistance")
        if pos.dtype not in (torch.float32, torch.float64):
            raise RuntimeError("torchmdnet_amd neighbour kernels are fp32 and fp64 (the reference's Warp kernels are instantiated for "
                               "these two, neighbors_brute.py:27); got " + str(pos.dtype))
        use_periodic = self.use_periodic or box is not None
        if use_periodic:
            if box is None:
                box = self.box
            box = box.to(device=pos.device, dtype=pos.dtype).contiguous()  # (no copy when the module lives on pos.device)
        else:
            box = None  # the zero buffer is never read: no host-to-device copy of it (not permitted while a stream is capturing)
        n = pos.shape[0]
        max_pairs = self.max_num_pairs if self.max_num_pairs >= 0 else -self.max_num_pairs * n
        if batch is None:
            batch = torch.zeros(n, dtype=torch.long, device=pos.device)
        batch = batch.to(torch.long).contiguous()
        # the molecule count is a host read-back; while the stream is being captured into a HIP graph (reference
        # tests/test_neighbors.py:452-588) the value of the warm-up calls is reused, as OutputModel.reduce does for dim_size
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self._n_mol = int(batch.max().item()) + 1 if n > 0 else 0
        elif getattr(self, "_n_mol", None) is None:
            raise RuntimeError("Warming up is needed before capturing OptimizedDistance into a CUDA graph")
        n_mol = self._n_mol
        # registered torch op (torchmdnet_amd/ops.py: fake impl + autograd = the reference's neighbor_grad_positions as a HIP
        # kernel), same outputs as torch.ops.torchmdnet.warp_neighbor_{brute,cell}_fwd (warp_ops/neighbors.py:34-148)
        from torchmdnet_amd import ops  # noqa: F401  (registers torch.ops.tmdnet.*)

        neighbors, deltas, dist, num_pairs = torch.ops.tmdnet.neighbor_pairs(
            pos, batch, box if use_periodic else None, float(self.cutoff_lower), float(self.cutoff_upper), int(max_pairs),
            bool(self.loop), bool(self.include_transpose), 1 if self.strategy == "cell" else 0, n_mol)
        if not torch.compiler.is_compiling() and not capturing and int(num_pairs.item()) > max_pairs:
            # reference: torch._assert_async -> RuntimeError (models/utils.py:297-300)
            raise RuntimeError("Found num_pairs > max_num_pairs, please increase max_num_pairs")
        edge_index, edge_vec, edge_weight = neighbors, deltas, dist
        if self.resize_to_fit:
            mask = edge_index[0] != -1
            edge_index = edge_index[:, mask]
            edge_weight = edge_weight[mask]
            edge_vec = edge_vec[mask, :]
        if not self.long_edge_index:
            edge_index = edge_index.to(torch.int32)
        return (edge_index, edge_weight, edge_vec) if self.return_vecs else (edge_index, edge_weight, None)
