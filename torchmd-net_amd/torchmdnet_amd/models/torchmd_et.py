"""Equivariant Transformer representation model: parameter containers with the reference's state-dict layout
(torchmdnet/models/torchmd_et.py:15-426, models/utils.py:45-117).

As for TensorNet, the reference's construction order and its nested ``reset_parameters`` calls are kept, so
``torch.manual_seed(s); create_model(args)`` yields bit-identical weights to the reference.  All arithmetic runs in
the HIP library (csrc/tn_et.hip, csrc/tn_et_api.hip).
"""
import torch
from torch import nn

from torchmdnet_amd.models.utils import CosineCutoff, OptimizedDistance, act_class_mapping, rbf_class_mapping

__all__ = ["TorchMD_ET"]


class NeighborEmbedding(nn.Module):
    """models/utils.py:45-81 (parameters only)."""

    def __init__(self, hidden_channels, num_rbf, cutoff_lower, cutoff_upper, max_z=100, dtype=torch.float32):
        super().__init__()
        self.embedding = nn.Embedding(max_z, hidden_channels, dtype=dtype)
        self.distance_proj = nn.Linear(num_rbf, hidden_channels, dtype=dtype)
        self.combine = nn.Linear(hidden_channels * 2, hidden_channels, dtype=dtype)
        self.cutoff = CosineCutoff(cutoff_lower, cutoff_upper)
        self.reset_parameters()

    def reset_parameters(self):
        self.embedding.reset_parameters()
        nn.init.xavier_uniform_(self.distance_proj.weight)
        nn.init.xavier_uniform_(self.combine.weight)
        self.distance_proj.bias.data.fill_(0)
        self.combine.bias.data.fill_(0)


class EquivariantMultiHeadAttention(nn.Module):
    """torchmd_et.py:238-313 (parameters only)."""

    def __init__(self, hidden_channels, num_rbf, distance_influence, num_heads, activation, attn_activation, cutoff_lower,
                 cutoff_upper, vector_cutoff=False, dtype=torch.float32):
        super().__init__()
        assert hidden_channels % num_heads == 0, (
            f"The number of hidden channels ({hidden_channels}) must be evenly divisible by the number of "
            f"attention heads ({num_heads})")
        self.distance_influence = distance_influence
        self.num_heads = num_heads
        self.hidden_channels = hidden_channels
        self.head_dim = hidden_channels // num_heads
        self.layernorm = nn.LayerNorm(hidden_channels, dtype=dtype)
        self.act = activation()
        self.attn_activation = act_class_mapping[attn_activation]()
        self.cutoff = CosineCutoff(cutoff_lower, cutoff_upper)
        self.q_proj = nn.Linear(hidden_channels, hidden_channels, dtype=dtype)
        self.k_proj = nn.Linear(hidden_channels, hidden_channels, dtype=dtype)
        self.v_proj = nn.Linear(hidden_channels, hidden_channels * 3, dtype=dtype)
        self.o_proj = nn.Linear(hidden_channels, hidden_channels * 3, dtype=dtype)
        self.vec_proj = nn.Linear(hidden_channels, hidden_channels * 3, bias=False, dtype=dtype)
        self.dk_proj = None
        if distance_influence in ["keys", "both"]:
            self.dk_proj = nn.Linear(num_rbf, hidden_channels, dtype=dtype)
        self.dv_proj = None
        if distance_influence in ["values", "both"]:
            self.dv_proj = nn.Linear(num_rbf, hidden_channels * 3, dtype=dtype)
        self.vector_cutoff = vector_cutoff
        self.reset_parameters()

    def reset_parameters(self):
        self.layernorm.reset_parameters()
        for lin in (self.q_proj, self.k_proj, self.v_proj, self.o_proj):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)
        nn.init.xavier_uniform_(self.vec_proj.weight)
        if self.dk_proj is not None:
            nn.init.xavier_uniform_(self.dk_proj.weight)
            self.dk_proj.bias.data.fill_(0)
        if self.dv_proj is not None:
            nn.init.xavier_uniform_(self.dv_proj.weight)
            self.dv_proj.bias.data.fill_(0)


class TorchMD_ET(nn.Module):
    """torchmd_et.py:15-186.  Supported on the HIP path: rbf_type "expnorm", activation / attn_activation "silu",
    cutoff_lower 0, head_dim a power of two <= 64."""

    def __init__(self, hidden_channels=128, num_layers=6, num_rbf=50, rbf_type="expnorm", trainable_rbf=True, activation="silu",
                 attn_activation="silu", neighbor_embedding=True, num_heads=8, distance_influence="both", cutoff_lower=0.0,
                 cutoff_upper=5.0, max_z=100, max_num_neighbors=32, box_vecs=None, vector_cutoff=False, static_shapes=False,
                 dtype=torch.float32):
        super().__init__()
        assert distance_influence in ["keys", "values", "both", "none"]
        assert rbf_type in rbf_class_mapping, f'Unknown RBF type "{rbf_type}". Choose from {", ".join(rbf_class_mapping.keys())}.'
        assert activation in act_class_mapping, f'Unknown activation function "{activation}".'
        assert attn_activation in act_class_mapping, f'Unknown attention activation function "{attn_activation}".'
        if activation != "silu" or attn_activation != "silu":
            raise NotImplementedError("the HIP Equivariant Transformer path implements SiLU activations (ET-SPICE.yaml)")
        hd = hidden_channels // num_heads
        if hidden_channels % num_heads or hd > 64 or (hd & (hd - 1)):
            raise NotImplementedError("head_dim must be a power of two <= 64 on the HIP path")
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.num_rbf = num_rbf
        self.rbf_type = rbf_type
        self.trainable_rbf = trainable_rbf
        self.activation = activation
        self.attn_activation = attn_activation
        self.num_heads = num_heads
        self.distance_influence = distance_influence
        self.cutoff_lower = cutoff_lower
        self.cutoff_upper = cutoff_upper
        self.max_z = max_z
        self.max_num_neighbors = max_num_neighbors
        self.vector_cutoff = vector_cutoff
        self.static_shapes = static_shapes
        self.dtype = dtype
        act_class = act_class_mapping[activation]
        self.embedding = nn.Embedding(self.max_z, hidden_channels, dtype=dtype)
        self.distance = OptimizedDistance(cutoff_lower, cutoff_upper, max_num_pairs=-max_num_neighbors, return_vecs=True, loop=True,
                                          box=box_vecs, long_edge_index=True)
        self.distance_expansion = rbf_class_mapping[rbf_type](cutoff_lower, cutoff_upper, num_rbf, trainable_rbf)
        self.neighbor_embedding = (NeighborEmbedding(hidden_channels, num_rbf, cutoff_lower, cutoff_upper, self.max_z, dtype)
                                   if neighbor_embedding else None)
        self.attention_layers = nn.ModuleList()
        for _ in range(num_layers):
            self.attention_layers.append(EquivariantMultiHeadAttention(
                hidden_channels, num_rbf, distance_influence, num_heads, act_class, attn_activation, cutoff_lower, cutoff_upper,
                vector_cutoff, dtype))
        self.out_norm = nn.LayerNorm(hidden_channels, dtype=dtype)
        self.reset_parameters()

    def reset_parameters(self):
        self.embedding.reset_parameters()
        self.distance_expansion.reset_parameters()
        if self.neighbor_embedding is not None:
            self.neighbor_embedding.reset_parameters()
        for attn in self.attention_layers:
            attn.reset_parameters()
        self.out_norm.reset_parameters()
