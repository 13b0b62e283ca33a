"""10k-atom periodic water box probe (BASELINE configs[4]-like, random-init weights; developer tool)."""
import os, sys, time, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W, _C
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
L = _C.lib()
for n_side in (15, 32):
    z, pos, box = W.water_box(n_side=n_side)
    m = create_model(dict(W.C2_ARGS, max_num_neighbors=96)).cuda()
    z, pos, box = z.cuda(), pos.cuda(), box.cuda()
    batch = torch.zeros_like(z)
    for _ in range(3): E, F = m(z, pos, batch, box=box)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): E, F = m(z, pos, batch, box=box)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{z.shape[0]} atoms periodic: {dt*1e3:.2f} ms/step -> {86400/dt*1e-6:.2f} ns/day @1fs, counts {m._engine.counts}, finite {torch.isfinite(F).all().item()}")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ncat = L.tmdnet_profile_num_categories()
    L.tmdnet_profile_begin(m._engine.handle, 0xFFFFFFFF)
    m(z, pos, batch, box=box)
    ms = (C.c_double * ncat)(); fl = (C.c_double * ncat)(); by = (C.c_double * ncat)(); ln = (C.c_int64 * ncat)()
    L.tmdnet_profile_end(m._engine.handle, sp, ms, fl, by, ln)
    print("   ", {L.tmdnet_profile_category_name(i).decode(): round(ms[i], 3) for i in range(ncat)})
