import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
for mnn in (64, 128):
  for nm in (1, 2, 8):
    m = create_model(dict(W.C2_ARGS)).cuda()
    z, pos, batch = W.synthetic_batch(n_mol=nm, n_atoms=64)
    z, pos, batch = z.cuda(), pos.cuda(), batch.cuda()
    E, F = m(z, pos, batch)
    ms = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=mnn)).cuda()
    ms.load_state_dict(m.state_dict())
    Es, Fs = ms(z, pos, batch)
    torch.cuda.synchronize()
    print(mnn, nm, "static eager vs dyn:", (Es - E).abs().max().item(), (Fs - F).abs().max().item(), ms._engine.counts, m._engine.counts)
    n = z.shape[0]
    for name, shape in [("phi", None), ("X_embed", (n, 9, 128)), ("X_layer0", (n, 9, 128)), ("X_layer1", (n, 9, 128)), ("x", (n, 128))]:
        if shape is None: continue
        a, b = ms.debug_tensor(name, shape), m.debug_tensor(name, shape)
        print("   ", name, (a - b).abs().max().item())
print("---- capture")
for nm in (1, 2, 8):
    m = create_model(dict(W.C2_ARGS)).cuda()
    z, pos, batch = W.synthetic_batch(n_mol=nm, n_atoms=64)
    z, pos, batch = z.cuda(), pos.cuda(), batch.cuda()
    E, F = m(z, pos, batch)
    ms = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=64)).cuda()
    ms.load_state_dict(m.state_dict())
    rp = ms.capture(z, pos, batch)
    for it in range(3):
        Eg, Fg = rp(pos)
        torch.cuda.synchronize()
        print(nm, it, "replay vs dyn:", (Eg.flatten() - E.flatten()).abs().max().item(), (Fg - F).abs().max().item(), ms.check_overflow(z.shape[0], nm))
