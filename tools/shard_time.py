import os,sys,time,torch
sys.path.insert(0,"torchmd-net_amd")
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import create_model
torch.manual_seed(0)
ms = create_model(dict(W.C2_ARGS, static_shapes=True, max_num_neighbors=64)).cuda()
out=[]
for nm in (16,32,64,128):
    z,pos,batch=(t.cuda() for t in W.synthetic_batch(n_mol=nm,n_atoms=64))
    r=ms.capture(z,pos,batch)
    for _ in range(20): r(pos)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): r(pos)
    torch.cuda.synchronize(); out.append((nm*64, round((time.perf_counter()-t0)/200*1e3,4)))
print(os.environ.get("TMDNET_EDGE_DIRECT_MAX"), out)
