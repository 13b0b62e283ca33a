#!/bin/bash
# developer probe: single-system HIP-graph latency per system size, fused small-system kernels on (default limit) and off
for mx in "" 0 ${EXTRA_MAX}; do
  TMDNET_SMALL_FUSED_MAX=$mx python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "torchmd-net_amd"))
if not os.environ.get("TMDNET_SMALL_FUSED_MAX"): os.environ.pop("TMDNET_SMALL_FUSED_MAX", None)
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
m = create_model(dict(W.C2_ARGS)).cuda()
for n in (64, 256, 512, 1024, 2048):
    sm = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=64)).cuda(); sm.load_state_dict(m.state_dict())
    z, pos, batch = (t.cuda() for t in W.synthetic_batch(n_mol=1, n_atoms=n))
    r = sm.capture(z, pos, batch)
    for _ in range(10): r(pos)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): r(pos)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print("fused_max", os.environ.get("TMDNET_SMALL_FUSED_MAX", "default"), n, "atoms", round(dt * 1e3, 4), "ms", round(86.4 / dt * 1e-3, 1), "ns/day")
PY
done
