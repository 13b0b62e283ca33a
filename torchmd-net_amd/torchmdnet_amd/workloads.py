"""Synthetic workloads shared by bench.py, the tests and the CPU baseline.

Nothing here touches the oracle or the HIP extension: it only builds inputs and the hyper-parameter
dicts of BASELINE.json's configs, so that the GPU path and the CPU baseline see identical data.
"""
import numpy as np
import torch

# BASELINE.json configs[1] ("C2"): the dict used by the reference's tests/test_model.py:167-187
C2_ARGS = dict(
    model="tensornet",
    embedding_dimension=128,
    num_layers=2,
    num_rbf=32,
    rbf_type="expnorm",
    trainable_rbf=False,
    activation="silu",
    cutoff_lower=0.0,
    cutoff_upper=5.0,
    max_z=100,
    max_num_neighbors=128,
    equivariance_invariance_group="O(3)",
    output_model="Scalar",
    reduce_op="sum",
    precision=32,
    prior_model=None,
    atom_filter=-1,
    derivative=True,
    static_shapes=False,
)

# a small configuration for fast parity tests (not a BASELINE config)
TINY_ARGS = dict(C2_ARGS, embedding_dimension=32, num_layers=2, num_rbf=16, max_z=20, max_num_neighbors=64)

# BASELINE.json configs[3] ("C4"): examples/ET-SPICE.yaml (Equivariant Transformer)
C4_ARGS = dict(
    model="equivariant-transformer",
    embedding_dimension=128,
    num_layers=5,
    num_heads=8,
    num_rbf=64,
    rbf_type="expnorm",
    trainable_rbf=False,
    activation="silu",
    attn_activation="silu",
    neighbor_embedding=True,
    distance_influence="both",
    vector_cutoff=True,
    cutoff_lower=0.0,
    cutoff_upper=10.0,
    max_z=100,
    max_num_neighbors=128,
    output_model="Scalar",
    reduce_op="sum",
    precision=32,
    prior_model=None,
    atom_filter=-1,
    derivative=True,
    static_shapes=False,
)
ET_TINY_ARGS = dict(C4_ARGS, embedding_dimension=32, num_layers=2, num_heads=4, num_rbf=16, max_z=20, max_num_neighbors=64,
                    cutoff_upper=5.0, vector_cutoff=False)

_Z_CHOICES = np.array([1, 6, 7, 8], dtype=np.int64)
_Z_PROBS = np.array([0.5, 0.3, 0.1, 0.1])


def synthetic_molecule(seed: int, n_atoms: int = 64, density: float = 0.1, min_dist: float = 0.9):
    """S-mol64 generator: ``n_atoms`` points uniform in a cube of side (n_atoms/density)^(1/3)
    with rejection of points closer than ``min_dist``; z ~ {H:.5, C:.3, N:.1, O:.1};
    RNG = numpy.default_rng(seed)."""
    rng = np.random.default_rng(seed)
    side = (n_atoms / density) ** (1.0 / 3.0)
    pts = np.empty((0, 3))
    while pts.shape[0] < n_atoms:
        cand = rng.uniform(0.0, side, size=3)
        if pts.shape[0] == 0 or np.min(np.linalg.norm(pts - cand, axis=1)) >= min_dist:
            pts = np.vstack([pts, cand])
    z = rng.choice(_Z_CHOICES, size=n_atoms, p=_Z_PROBS)
    return z, pts.astype(np.float32)


def synthetic_batch(n_mol: int = 256, n_atoms: int = 64, first_seed: int = 0, device="cpu"):
    """``n_mol`` independent molecules, molecule m from seed ``first_seed + m``;
    returns (z int64 [N], pos float32 [N,3], batch int64 [N])."""
    zs, ps = [], []
    for m in range(n_mol):
        z, p = synthetic_molecule(first_seed + m, n_atoms)
        zs.append(z)
        ps.append(p)
    z = torch.from_numpy(np.concatenate(zs))
    pos = torch.from_numpy(np.concatenate(ps))
    batch = torch.repeat_interleave(torch.arange(n_mol, dtype=torch.long), n_atoms)
    return z.to(device), pos.to(device), batch.to(device)


def water_box(n_side: int = 15, spacing: float = 3.1, seed: int = 0, jitter: float = 0.3):
    """S-water-like periodic box: n_side^3 rigid-ish waters on a jittered cubic lattice.
    Returns (z, pos, box[3,3]).  Used for the single-system / periodic cases."""
    rng = np.random.default_rng(seed)
    L = n_side * spacing
    g = np.arange(n_side) * spacing + 0.5 * spacing
    ox = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    ox = ox + rng.uniform(-jitter, jitter, size=ox.shape)
    n = ox.shape[0]
    # random orientation of an O-H 0.96 A, 104.5 deg water
    def rand_unit(k):
        v = rng.normal(size=(k, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    a = rand_unit(n)
    b = rand_unit(n)
    b = b - (b * a).sum(1, keepdims=True) * a
    b = b / np.linalg.norm(b, axis=1, keepdims=True)
    half = np.deg2rad(104.5 / 2)
    h1 = ox + 0.96 * (np.cos(half) * a + np.sin(half) * b)
    h2 = ox + 0.96 * (np.cos(half) * a - np.sin(half) * b)
    pos = np.stack([ox, h1, h2], 1).reshape(-1, 3).astype(np.float32)
    z = np.tile(np.array([8, 1, 1], dtype=np.int64), n)
    box = np.diag([L, L, L]).astype(np.float32)
    return torch.from_numpy(z), torch.from_numpy(pos), torch.from_numpy(box)
