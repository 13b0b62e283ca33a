"""TensorNet2 + ScalarPlusWeightedCoulomb at the bench's size (256 x 64 atoms): ms per step and per-class times (developer tool)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch, bench
from torchmdnet_amd import _C
r = bench.tn2_leg(torch.device("cuda"), _C.lib(), steps=10, warmup=3)
print(json.dumps({"ms_per_step": r["ms_per_step"], "classes_ms": r["classes_ms"]}))
g = bench.AUX_GROUPS.get("tensornet2", {})
for k, v in sorted(g.items(), key=lambda kv: -kv[1]["ms"])[:9]:
    print(f"  {v['ms']:.3f} x{v['launches']}  {k[:80]}")
