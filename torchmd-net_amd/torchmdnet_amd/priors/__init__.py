from torchmdnet_amd.priors.atomref import Atomref, BasePrior

__all__ = ["Atomref"]
