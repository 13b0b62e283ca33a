"""Output heads (reference torchmdnet/models/output_modules.py).  Only ``Scalar`` is on the energy+force
path; it is a parameter container here, the MLP and the per-molecule reduction run in the HIP library."""
import torch
from torch import nn

from torchmdnet_amd.models.utils import MLP

# the reference's own list (output_modules.py:19), which its tests parametrise over: of these only "Scalar" is in SURVEY section 8's
# scope - the two property heads are defined below as classes that raise NotImplementedError on construction, so that
# `from ...output_modules import *` works and create_model refuses them by name.  What the engine evaluates: __engine_heads__
__all__ = ["Scalar", "DipoleMoment", "ElectronicSpatialExtent"]
__engine_heads__ = ["Scalar", "EquivariantScalar", "ScalarPlusWeightedCoulomb"]


class OutputModel(nn.Module):
    def __init__(self, allow_prior_model, reduce_op, static_shapes=False):
        super().__init__()
        self.allow_prior_model = allow_prior_model
        self.reduce_op = reduce_op
        self.static_shapes = static_shapes
        self.dim_size = 0

    def reset_parameters(self):
        pass


class Scalar(OutputModel):
    """F -> F/2 -> 1 per atom (output_modules.py:79-117)."""

    def __init__(self, hidden_channels, activation="silu", allow_prior_model=True, reduce_op="sum", dtype=torch.float,
                 static_shapes=False, **kwargs):
        super().__init__(allow_prior_model=allow_prior_model, reduce_op=reduce_op, static_shapes=static_shapes)
        if kwargs.get("num_layers", 0) != 0 or kwargs.get("num_hidden_layers", 0) != 0:
            raise NotImplementedError("Scalar head with extra hidden layers has no HIP kernel yet")
        if reduce_op not in ("sum", "add"):
            raise NotImplementedError(f"reduce_op={reduce_op} has no HIP kernel (sum/add only)")
        self.output_network = MLP(in_channels=hidden_channels, out_channels=1, hidden_channels=hidden_channels // 2,
                                  activation=activation, num_hidden_layers=0, dtype=dtype)
        self.reset_parameters()

    def reset_parameters(self):
        self.output_network.reset_parameters()


class GatedEquivariantBlock(nn.Module):
    """models/utils.py:583-625 (parameters only)."""

    def __init__(self, hidden_channels, out_channels, intermediate_channels=None, activation="silu", scalar_activation=False,
                 dtype=torch.float):
        super().__init__()
        self.out_channels = out_channels
        if intermediate_channels is None:
            intermediate_channels = hidden_channels
        self.vec1_proj = nn.Linear(hidden_channels, hidden_channels, bias=False, dtype=dtype)
        self.vec2_proj = nn.Linear(hidden_channels, out_channels, bias=False, dtype=dtype)
        self.update_net = MLP(in_channels=hidden_channels * 2, out_channels=out_channels * 2, hidden_channels=intermediate_channels,
                              activation=activation, num_hidden_layers=0, dtype=dtype)
        self.scalar_activation = scalar_activation

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.vec1_proj.weight)
        nn.init.xavier_uniform_(self.vec2_proj.weight)
        self.update_net.reset_parameters()


class EquivariantScalar(OutputModel):
    """Two gated equivariant blocks F -> F/2 -> 1 (output_modules.py:120-163); what create_model builds for
    output_model="Scalar" on the Equivariant Transformer (model.py:134-135)."""

    def __init__(self, hidden_channels, activation="silu", allow_prior_model=True, reduce_op="sum", dtype=torch.float,
                 static_shapes=False, **kwargs):
        super().__init__(allow_prior_model=allow_prior_model, reduce_op=reduce_op, static_shapes=static_shapes)
        if activation != "silu":
            raise NotImplementedError("the HIP EquivariantScalar head implements SiLU")
        if reduce_op not in ("sum", "add"):
            raise NotImplementedError(f"reduce_op={reduce_op} has no HIP kernel (sum/add only)")
        self.output_network = nn.ModuleList([
            GatedEquivariantBlock(hidden_channels, hidden_channels // 2, activation=activation, scalar_activation=True, dtype=dtype),
            GatedEquivariantBlock(hidden_channels // 2, 1, activation=activation, dtype=dtype),
        ])
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.output_network:
            layer.reset_parameters()


class _BoxHolder(nn.Module):
    """state-dict slot ``output_model.distance.box`` of the reference's cutoff-mode Coulomb head (output_modules.py:424-437)"""

    def __init__(self):
        super().__init__()
        self.register_buffer("box", torch.zeros((3, 3)), persistent=True)


class ScalarPlusWeightedCoulomb(OutputModel):
    """Scalar head + damped pair Coulomb energy of the predicted charge channels (output_modules.py:344-441): parameter
    container (``output_network`` MLP F -> F/2 -> 1, buffer ``qweights`` [(num_layers + 1) * q_dim]); the pair sum (all pairs
    of a molecule, or the reaction field inside ``coulomb_cutoff``) runs in the HIP library."""

    def __init__(self, hidden_channels, activation="silu", allow_prior_model=True, reduce_op="sum", dtype=torch.float,
                 static_shapes=False, cutoff=None, **kwargs):
        super().__init__(allow_prior_model=allow_prior_model, reduce_op=reduce_op, static_shapes=static_shapes)
        if kwargs.get("num_hidden_layers", 0) != 0:
            raise NotImplementedError("ScalarPlusWeightedCoulomb head with extra hidden layers has no HIP kernel yet")
        if reduce_op not in ("sum", "add"):
            raise NotImplementedError(f"reduce_op={reduce_op} has no HIP kernel (sum/add only)")
        self.hidden_channels = hidden_channels
        self.output_network = MLP(in_channels=hidden_channels, out_channels=1, hidden_channels=hidden_channels // 2,
                                  activation=activation, num_hidden_layers=0, dtype=dtype)
        self.q_dim = kwargs["q_dim"]
        self.num_interaction_layers = kwargs["num_layers"]
        self.layer_weights = kwargs["q_weights"]
        self.cutoff = kwargs["coulomb_cutoff"]
        assert len(self.layer_weights) == self.num_interaction_layers + 1
        w = torch.zeros((self.num_interaction_layers + 1, self.q_dim), dtype=dtype)
        for i in range(self.num_interaction_layers + 1):
            w[i, :] = torch.tensor(self.layer_weights[i], dtype=dtype)
        self.register_buffer("qweights", w.flatten())
        self.mode = "all_to_all" if self.cutoff is None else "cutoff"
        self.epsilon_solvent = kwargs.get("coulomb_epsilon_solvent", 78.3)
        if self.cutoff is not None:  # the reference keeps a neighbour module here: its `box` buffer is part of the state dict
            self.distance = _BoxHolder()
        self.reset_parameters()

    def reset_parameters(self):
        self.output_network.reset_parameters()


class _OutOfScopeHead(OutputModel):
    """Property heads of the reference (output_modules.py:166-341) outside SURVEY section 8's energy + force path."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"output_model={type(self).__name__} is outside the HIP energy+force path (Scalar, EquivariantScalar, "
                                  "ScalarPlusWeightedCoulomb)")


class DipoleMoment(_OutOfScopeHead):
    pass


class ElectronicSpatialExtent(_OutOfScopeHead):
    pass
