"""Stand-ins for tests/example.ckpt (Equivariant Transformer) and tests/example_tensornet.ckpt, which the reference's tests load
and the reference checkout itself lacks (/root/reference/.MISSING_LARGE_BLOBS).  Written in the reference's Lightning checkpoint
layout (`hyper_parameters` + `state_dict` with the `model.` prefix, reference model.py:208-374): random-init weights (seed 1234) of
the example YAMLs' architectures.  The tests that load them compare the model with ITSELF through another entry point (ensemble of
three copies, External adapter vs direct call), so any valid checkpoint exercises them; their known-answer variants (caffeine /
AceFF energies) stay unpinned (DESIGN: real-checkpoint parity)."""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
from torchmdnet_amd.models.model import create_model  # noqa: E402

out_dir = sys.argv[1]
examples = os.path.join(os.path.dirname(out_dir), "examples")
for fname, yml, name in (("example.ckpt", "ET-QM9.yaml", "equivariant-transformer"), ("example_tensornet.ckpt", "TensorNet-QM9.yaml", "tensornet")):
    args = yaml.load(open(os.path.join(examples, yml)), Loader=yaml.FullLoader)
    args.update(model=name, prior_model=None, precision=32, box_vecs=None, remove_ref_energy=False, derivative=False)
    torch.manual_seed(1234)
    model = create_model(args)
    sd = {"model." + k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.save({"hyper_parameters": args, "state_dict": sd}, os.path.join(out_dir, fname))
    print("wrote", fname, len(sd), "tensors")
