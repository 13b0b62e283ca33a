"""-m gpu: the forces are the negative gradient of the energy along a trajectory.  Velocity-Verlet NVE on the captured HIP
graph (static shapes, neighbour list rebuilt inside the graph every step - the MD mode of BASELINE configs[4]): with
F = -dE/dx the total energy is conserved up to O(dt^2) fluctuations and the fluctuation drops ~4x when dt is halved; a force
that is inconsistent with the energy (a wrong adjoint, a stale neighbour list, a replay that skips a kernel) shows as drift.
Random-init weights: an arbitrary but smooth potential."""
import pytest
import torch

from torchmdnet_amd import workloads as W

pytestmark = pytest.mark.gpu


def _nve(replay, pos0, vel0, dt, steps, mass=12.0):
    pos, vel = pos0.clone(), vel0.clone()
    e, f = replay(pos)
    f = f.clone()
    tot = []
    for _ in range(steps):
        vel = vel + 0.5 * dt * f / mass
        pos = pos + dt * vel
        e, f = replay(pos)
        f = f.clone()
        vel = vel + 0.5 * dt * f / mass
        tot.append(float(e.sum()) + 0.5 * mass * float((vel * vel).sum()))
    return torch.tensor(tot, dtype=torch.float64), pos


@pytest.mark.parametrize("arch", ["tensornet", "equivariant-transformer", "tensornet2"])
def test_nve_energy_conservation_on_graph_replay(hip_lib, arch):
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(4)
    if arch == "tensornet":
        args = dict(W.TINY_ARGS, static_shapes=True)
    elif arch == "equivariant-transformer":
        args = dict(W.ET_TINY_ARGS, static_shapes=True)
    else:
        args = dict(W.TINY_ARGS, static_shapes=True, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=4,
                    q_weights=[1.0, 1.0, 1.0])
    model = create_model(args).to("cuda")
    z, pos, batch = W.synthetic_batch(n_mol=1, n_atoms=40, first_seed=31)
    z, pos, batch = (z % 8 + 1).cuda(), pos.cuda(), batch.cuda()
    q = torch.zeros(1, device="cuda") if arch != "equivariant-transformer" else None
    replay = model.capture(z, pos, batch, q=q)
    vel = 0.02 * torch.randn(pos.shape, generator=torch.Generator().manual_seed(8)).cuda()
    e0, f0 = replay(pos)
    fscale = float(f0.abs().max())
    dt = min(0.05, 0.02 * (12.0 / max(fscale, 1e-6)) ** 0.5)  # small against the stiffest force on the lightest scale
    # 120 steps: long enough for the atoms to move by ~0.2 A, short enough that the (random-init, partly attractive)
    # potential has not yet pulled atoms into contact, where any integrator loses the energy
    tot1, end1 = _nve(replay, pos, vel, dt, 120)
    tot2, end2 = _nve(replay, pos, vel, dt / 2, 240)
    kin = 0.5 * 12.0 * float((vel * vel).sum())
    fl1, fl2 = float(tot1.std()), float(tot2.std())
    drift1 = abs(float(tot1[-20:].mean() - tot1[:20].mean()))
    assert (end1 - pos).abs().max().item() > 10 * dt * 0.02  # the atoms really moved
    assert drift1 < 0.02 * max(kin, float(tot1.abs().mean()) * 1e-3), (drift1, kin)
    assert fl2 < 0.6 * fl1 + 1e-7 * abs(float(tot1.mean())), (fl1, fl2)  # O(dt^2): ~0.25x, fp32 noise floor aside
