"""Atomref prior: per-atom reference energy table added before the molecule sum
(reference torchmdnet/priors/atomref.py:12-96, base.py:26-63).  Needed to consume AceFF-1.x
checkpoints (load_model toggles ``enable``, models/model.py:250-259).  The table is handed to the HIP
head kernel; nothing is evaluated in Python."""
import torch
from torch import nn


class BasePrior(nn.Module):
    def __init__(self, dataset=None):
        super().__init__()

    def get_init_args(self):
        return {}


class Atomref(BasePrior):
    def __init__(self, max_z=None, dataset=None, trainable=False, enable=True):
        super().__init__()
        if max_z is None and dataset is None:
            raise ValueError("Can't instantiate Atomref prior, all arguments are None.")
        if dataset is None:
            atomref = torch.zeros(max_z, 1)
        else:
            atomref = dataset.get_atomref()
            if atomref is None:
                atomref = torch.zeros(100, 1)
        if atomref.ndim == 1:
            atomref = atomref.view(-1, 1)
        self.register_buffer("initial_atomref", atomref)
        self.atomref = nn.Embedding(len(atomref), 1, _freeze=not trainable, _weight=atomref)
        self.enable = enable

    def reset_parameters(self):
        self.atomref.weight.data.copy_(self.initial_atomref)

    def get_init_args(self):
        return dict(max_z=self.initial_atomref.size(0), trainable=self.atomref.weight.requires_grad, enable=self.enable)
