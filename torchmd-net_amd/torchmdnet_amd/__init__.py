"""torchmdnet_amd -- host-side mirror of the torchmd-net energy+force surface over the MI355X-native
HIP engine (libtmdnet_amd.so, include/tmdnet_amd.h).

Public entry points (same names and semantics as the reference's torchmdnet.models.model):
    create_model, load_model, load_ensemble, TorchMD_Net, Ensemble
`install_as_torchmdnet()` aliases this package as ``torchmdnet`` so that code written against the
reference (tests/test_model.py, TMDNETCalculator, OpenMM wrappers) imports it unchanged.
"""
import sys

__version__ = "0.1.0"


def install_as_torchmdnet():
    """Register ``torchmdnet``, ``torchmdnet.models``, ``torchmdnet.models.model`` ... aliases."""
    import importlib

    names = ["", ".models", ".models.model", ".models.utils", ".models.tensornet", ".models.torchmd_et",
             ".models.output_modules", ".priors"]
    for n in names:
        mod = importlib.import_module("torchmdnet_amd" + n)
        sys.modules["torchmdnet" + n] = mod
