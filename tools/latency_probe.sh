#!/bin/bash
# developer probe: single-system HIP-graph latency per system size; per-atom phase kernels on (defaults: one atom per block up to
# 512 atoms, four atoms per block up to 4 096) and off
for cfg in "on" "off"; do
  if [ "$cfg" = "off" ]; then export TMDNET_SMALL_FUSED_MAX=0 TMDNET_MID_FUSED_MAX=0; fi
  CFG=$cfg python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
m = create_model(dict(W.C2_ARGS)).cuda()
for n_mol, n in [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "1x64,1x256,1x512,16x64,32x64,64x64").split(",")]:
    sm = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=64)).cuda(); sm.load_state_dict(m.state_dict())
    z, pos, batch = (t.cuda() for t in W.synthetic_batch(n_mol=n_mol, n_atoms=n))
    r = sm.capture(z, pos, batch)
    for _ in range(10): r(pos)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): r(pos)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("phase kernels", os.environ["CFG"], f"{n_mol} x {n} = {n_mol * n} atoms", round(dt * 1e3, 4), "ms")
PY
done
