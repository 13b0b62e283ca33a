"""One-off timing of the UNMODIFIED reference (torchmd-net, pure-PyTorch TensorNet path, autograd forces) on the BUILD
container's CPU, BASELINE.md section 3 protocol: model = create_model(C2 dict) after torch.manual_seed(0), S-mol64 batch
in chunks of 16 molecules (the naive neighbour shim enumerates all atom pairs of a chunk; chunks keep that O(N^2) part
and the [E,3,3,F] temporaries small), 2 warm-ups + best of 5 timed `model(z, pos, batch)` calls per chunk, chunk times
summed.  The reference cannot travel to the GPU box, so this figure is committed (profiles/r02_reference_cpu.json) and
quoted beside bench.py's cpu_baseline (kind "port", timed on the GPU host).  Also times the port on the same chunks: the
ratio says how faithful the port is as a stand-in.   python tools/time_reference_cpu.py [n_chunks]"""
import json, os, platform, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from oracle import ref_shims as R, tensornet_torch as T
from torchmdnet_amd import workloads as W

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chunk = 16
mm = R.reference_model_module()
torch.manual_seed(0)
model = mm.create_model(dict(W.C2_ARGS))
model.eval()
sd = {k: v.detach() for k, v in model.state_dict().items()}
hp = T.hparams_from_args(W.C2_ARGS)
tot_ref = tot_port = 0.0
for c in range(n_chunks):
    z, pos, batch = W.synthetic_batch(n_mol=chunk, first_seed=c * chunk)
    tr, tp = [], []
    for rep in range(7):
        t0 = time.perf_counter(); E, F = model(z, pos.clone(), batch); t1 = time.perf_counter()
        Ep, Fp = T.energy_and_forces(sd, hp, z, pos, batch); t2 = time.perf_counter()
        if rep >= 2:
            tr.append(t1 - t0); tp.append(t2 - t1)
    assert torch.allclose(E.detach(), Ep, atol=1e-4, rtol=1e-4) and torch.allclose(F.detach(), Fp, atol=1e-4, rtol=1e-4)
    tot_ref += min(tr); tot_port += min(tp)
    print(f"chunk {c}: reference {min(tr):.3f} s, port {min(tp):.3f} s", flush=True)
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
out = {"what": "unmodified reference (oracle/ref_shims.py) vs oracle/tensornet_torch.py, TensorNet C2, E+F, fp32, CPU",
       "molecules": n_chunks * chunk, "chunk": chunk, "reference_molecules_per_s": n_chunks * chunk / tot_ref,
       "port_molecules_per_s": n_chunks * chunk / tot_port, "threads": torch.get_num_threads(), "cores": os.cpu_count(),
       "cpu": cpu[0] if cpu else platform.processor(), "torch": torch.__version__, "protocol": "2 warm-ups + best of 5 per chunk"}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json"), "w"), indent=1)
