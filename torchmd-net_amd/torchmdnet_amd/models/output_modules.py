"""Output heads (reference torchmdnet/models/output_modules.py).  Only ``Scalar`` is on the energy+force
path; it is a parameter container here, the MLP and the per-molecule reduction run in the HIP library."""
import torch
from torch import nn

from torchmdnet_amd.models.utils import MLP

__all__ = ["Scalar"]


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
