"""torchmdnet_amd -- host-side mirror of the torchmd-net energy+force surface over the MI355X-native
HIP engine (libtmdnet_amd.so, include/tmdnet_amd.h).

Public entry points (same names and semantics as the reference's torchmdnet.models.model):
    create_model, load_model, load_ensemble, TorchMD_Net, Ensemble
`install_as_torchmdnet()` aliases this package as ``torchmdnet`` so that code written against the
reference (tests/test_model.py, TMDNETCalculator, OpenMM wrappers) imports it unchanged.
"""
import sys

__version__ = "0.3.0"


def install_as_torchmdnet(reference_root=None):
    """Register ``torchmdnet``, ``torchmdnet.models``, ``torchmdnet.models.model`` ... aliases of this package.

    The reference's adapters (``torchmdnet/calculators.py``: ``External`` for TorchMD, ``TMDNETCalculator`` for ASE) are
    NOT re-implemented here: they contain no arithmetic, only ``load_model`` + ``model(z, pos, batch, box)`` calls.  When
    ``reference_root`` (or ``$TMDNET_REFERENCE_ROOT``) points at a checkout / install of the reference, its
    ``calculators.py`` is loaded by path under the name ``torchmdnet.calculators`` AFTER the aliases are in place, so its
    ``from torchmdnet.models.model import load_model`` binds to the HIP engine.  Returns the list of aliased module names."""
    import importlib
    import importlib.util
    import os

    names = ["", ".models", ".models.model", ".models.utils", ".models.tensornet", ".models.tensornet2", ".models.torchmd_et",
             ".models.output_modules", ".priors", ".ops"]
    done = []
    for n in names:
        mod = importlib.import_module("torchmdnet_amd" + n)
        sys.modules["torchmdnet" + n] = mod
        done.append("torchmdnet" + n)
    root = reference_root or os.environ.get("TMDNET_REFERENCE_ROOT")
    if root:
        path = os.path.join(root, "torchmdnet", "calculators.py")
        if os.path.isfile(path):
            spec = importlib.util.spec_from_file_location("torchmdnet.calculators", path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["torchmdnet.calculators"] = mod
            try:
                spec.loader.exec_module(mod)  # needs `ase` (module-level import of the ASE base class)
                done.append("torchmdnet.calculators")
            except ImportError:
                del sys.modules["torchmdnet.calculators"]
                raise
    return done
