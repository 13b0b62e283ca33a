"""Kernel-level view of the single-system latency path (run under rocprofv3 --kernel-trace --stats): one 64-atom
molecule, static shapes, HIP-graph replay (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
ms = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=64)).cuda()
z, pos, batch = W.synthetic_batch(n_mol=int(os.environ.get("N_MOL", "1")), n_atoms=int(os.environ.get("N_ATOMS", "64")))
z, pos, batch = z.cuda(), pos.cuda(), batch.cuda()
replay = ms.capture(z, pos, batch)
for _ in range(200): replay(pos)
torch.cuda.synchronize()
