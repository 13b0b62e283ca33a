"""TensorNet2 representation model (AceFF-2.0): parameter containers with the reference's state-dict layout
(torchmdnet/models/tensornet2.py:49-66 ChargePredict, :228-330 TensorNet2, :466-520 Interaction).

TensorNet with neutral charge equilibration: a ChargePredict head after the embedding and after every interaction layer
(LayerNorm(3F) -> MLP 3F -> F -> F -> 2 q_dim -> per-molecule equilibration to the total charge), and interaction layers
whose edge MLP takes [phi(d), c_i, c_j].  Construction order and ``reset_parameters`` sequence follow the reference, so
``torch.manual_seed(s); create_model(args)`` gives the reference's weights.  All arithmetic runs in the HIP library.
"""
import torch
from torch import nn

from torchmdnet_amd.models.tensornet import TensorEmbedding
from torchmdnet_amd.models.utils import MLP, CosineCutoff, OptimizedDistance, act_class_mapping, rbf_class_mapping

__all__ = ["TensorNet2"]


class ChargePredict(nn.Module):
    def __init__(self, hidden_channels, activation, q_dim=16, static_shapes=False):
        super().__init__()
        self.q_dim = q_dim
        self.q_norm = nn.LayerNorm(3 * hidden_channels)
        self.q_mlp = MLP(3 * hidden_channels, 2 * q_dim, hidden_channels, activation, 1)
        self.static_shapes = static_shapes
        self.dim_size = 0

    def reset_parameters(self):
        self.q_norm.reset_parameters()
        self.q_mlp.reset_parameters()


class Interaction(nn.Module):
    def __init__(self, num_rbf, hidden_channels, q_dim, activation, cutoff_lower, cutoff_upper, equivariance_invariance_group,
                 dtype=torch.float32):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.cutoff = CosineCutoff(cutoff_lower, cutoff_upper)
        self.linears_scalar = nn.ModuleList(
            [nn.Linear(num_rbf + 2 * q_dim, hidden_channels, bias=True, dtype=dtype),
             nn.Linear(hidden_channels, 2 * hidden_channels, bias=True, dtype=dtype),
             nn.Linear(2 * hidden_channels, 3 * hidden_channels, bias=True, dtype=dtype)])
        self.linears_tensor = nn.ModuleList(nn.Linear(hidden_channels, hidden_channels, bias=False) for _ in range(6))
        self.act = act_class_mapping[activation]()
        self.equivariance_invariance_group = equivariance_invariance_group
        self.reset_parameters()

    def reset_parameters(self):
        for lin in self.linears_scalar:
            lin.reset_parameters()
        for lin in self.linears_tensor:
            lin.reset_parameters()


class TensorNet2(nn.Module):
    """Same constructor signature and attributes as the reference class (tensornet2.py:228-330)."""

    def __init__(self, hidden_channels=128, q_dim=16, num_layers=2, num_rbf=32, rbf_type="expnorm", trainable_rbf=False,
                 activation="silu", cutoff_lower=0, cutoff_upper=4.5, max_num_neighbors=64, max_z=128,
                 equivariance_invariance_group="O(3)", static_shapes=True, dtype=torch.float32, box_vecs=None,
                 output_charges=False):
        super().__init__()
        if rbf_type not in rbf_class_mapping:
            raise NotImplementedError(f'RBF type "{rbf_type}" has no HIP kernel (available: {list(rbf_class_mapping)})')
        if activation not in act_class_mapping:
            raise NotImplementedError(f'activation "{activation}" has no HIP kernel (available: {list(act_class_mapping)})')
        assert equivariance_invariance_group in ["O(3)", "SO(3)"], (
            f'Unknown group "{equivariance_invariance_group}". Choose O(3) or SO(3).')
        if not output_charges:
            raise NotImplementedError("TensorNet2 has a HIP path with the ScalarPlusWeightedCoulomb head only (AceFF-2.0)")
        if q_dim <= 0 or (num_layers + 1) * q_dim > 64:
            raise NotImplementedError("the HIP Coulomb kernel takes 1 <= (num_layers + 1) * q_dim <= 64 charge channels")
        self.q_dim = q_dim
        self.output_charges = output_charges
        self.hidden_channels = hidden_channels
        self.equivariance_invariance_group = equivariance_invariance_group
        self.num_layers = num_layers
        self.num_rbf = num_rbf
        self.rbf_type = rbf_type
        self.activation = activation
        self.cutoff_lower = cutoff_lower
        self.cutoff_upper = cutoff_upper
        self.max_z = max_z
        self.max_num_neighbors = max_num_neighbors
        act_class = act_class_mapping[activation]
        self.distance_expansion = rbf_class_mapping[rbf_type](cutoff_lower, cutoff_upper, num_rbf, trainable_rbf)
        self.tensor_embedding = TensorEmbedding(hidden_channels, num_rbf, act_class, cutoff_lower, cutoff_upper,
                                                trainable_rbf, max_z, dtype)
        self.charge_predict_0 = ChargePredict(hidden_channels, activation, q_dim, static_shapes)
        self.layers = nn.ModuleList()
        self.charge_predicts = nn.ModuleList()
        for _ in range(num_layers):
            self.layers.append(Interaction(num_rbf, hidden_channels, q_dim, activation, cutoff_lower, cutoff_upper,
                                           equivariance_invariance_group, dtype))
            self.charge_predicts.append(ChargePredict(hidden_channels, activation, q_dim, static_shapes))
        self.linear = nn.Linear(3 * hidden_channels, hidden_channels, dtype=dtype)
        self.out_norm = nn.LayerNorm(3 * hidden_channels, dtype=dtype)
        self.act = act_class()
        self.static_shapes = static_shapes
        self.distance = OptimizedDistance(cutoff_lower, cutoff_upper, max_num_pairs=-max_num_neighbors, return_vecs=True,
                                          loop=True, resize_to_fit=not self.static_shapes, box=box_vecs, long_edge_index=True)
        self.opt = True
        self.reset_parameters()

    def reset_parameters(self):
        self.tensor_embedding.reset_parameters()
        for layer in self.layers:
            layer.reset_parameters()
        self.linear.reset_parameters()
        self.out_norm.reset_parameters()
        self.charge_predict_0.reset_parameters()
        for layer in self.charge_predicts:
            layer.reset_parameters()

    def forward(self, *args, **kwargs):
        raise RuntimeError("torchmdnet_amd.TensorNet2 is evaluated through TorchMD_Net.forward (fused HIP schedule); "
                           "the representation alone is not exposed as a torch graph")
