"""Times of the analytic second-order pass per architecture on the bench batch (developer tool): TensorNet (C2), the Equivariant
Transformer (ET-SPICE hyper-parameters) and TensorNet2 + Coulomb head, 256 x 64 atoms each:
  pass_ms           tmdnet_force_param_grads alone (parameters only)
  pass_hv_ms        with H v
  pass_seeded_ms    tmdnet_loss_param_grads: with H v and the energy seed (one-pass training)
  first_order_ms    tmdnet_energy_param_grads (what the seeded pass saves)
  step_*_ms         loss(E, F).backward() + SGD step: one pass / two passes / order-2 difference quotient
Writes gpurun_out/second_order_time.json.  The numbers decide `force_gradient_order`'s automatic choice per architecture.
usage: python tools/second_order_time.py [tensornet] [et] [tn2]      (default: all three)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from torchmdnet_amd import workloads as W  # noqa: E402
from torchmdnet_amd.models.model import create_model  # noqa: E402

ARGS = {
    "tensornet": dict(W.C2_ARGS),
    "et": dict(W.C4_ARGS),
    "tn2": dict(W.C2_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=16, q_weights=[1.0, 1.0, 1.0]),
}


def timed(fn, reps=4, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    which = [a for a in sys.argv[1:] if a in ARGS] or list(ARGS)
    z, pos, batch = (t.to(dev) for t in W.synthetic_batch(n_mol=bench.N_MOL, n_atoms=bench.N_ATOMS))
    B = bench.N_MOL
    v = torch.randn_like(pos)
    ge = torch.randn(B, device=dev)
    out = {}
    for arch in which:
        r = {}
        try:
            torch.manual_seed(0)
            model = create_model(dict(ARGS[arch], derivative=True)).to(dev)
            model.parameter_gradients = True
            q = torch.zeros(B, device=dev) if arch == "tn2" else None
            r["inference_ms"] = timed(lambda: model.energy_and_forces(z, pos, batch, None, q, B))
            r["first_order_ms"] = timed(lambda: model.parameter_gradients_of(z, pos, batch, None, q, B, ge))
            r["pass_ms"] = timed(lambda: model.force_term_parameter_gradients(z, pos, batch, None, q, B, v))
            r["pass_hv_ms"] = timed(lambda: model.force_term_parameter_gradients(z, pos, batch, None, q, B, v, want_hv=True))
            r["pass_seeded_ms"] = timed(lambda: model.force_term_parameter_gradients(z, pos, batch, None, q, B, v, want_hv=True, ge=ge))
            r["workspace_GiB"] = model._engine.hvp_ws.numel() / 2 ** 30
            opt = torch.optim.SGD(model.parameters(), lr=1e-7)

            def step():
                opt.zero_grad()
                y, f = model(z, pos, batch, q=q)
                ((y ** 2).mean() + (f ** 2).mean()).backward()
                opt.step()

            for key, order, one in (("step_one_pass_ms", 0, True), ("step_two_passes_ms", 0, False), ("step_difference_quotient_ms", 2, False)):
                model.force_gradient_order, model.one_pass_training = order, one
                r[key] = timed(step, reps=3, warm=2)
        except Exception as e:  # keep the other architectures' numbers
            r["error"] = f"{type(e).__name__}: {e}"
        out[arch] = r
        print(arch, json.dumps(r), flush=True)
        del model
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "second_order_time.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
