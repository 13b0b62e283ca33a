"""-m gpu: the HIP neighbour list behind OptimizedDistance vs the numpy specification
(oracle/neighbors_numpy.py = reference tests/test_neighbors.py:14-71).  Integer output (pair set) is
compared exactly; distances / vectors to np.allclose defaults like the reference test (:142-148)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _system(n_batches, dtype=torch.float32, lbox=10.0, seed=4321):
    torch.manual_seed(seed)
    n_per = torch.randint(3, 100, size=(n_batches,))
    batch = torch.repeat_interleave(torch.arange(n_batches, dtype=torch.int64), n_per)
    pos = torch.rand(int(n_per.sum()), 3, dtype=dtype) * lbox - 10.0 * lbox
    pos[0, :] = 0
    pos[1, :] = 0  # at least one pair (coincident atoms: d = 0 between distinct atoms)
    return pos, batch


@pytest.mark.parametrize("strategy", ["brute", "cell"])
@pytest.mark.parametrize("n_batches", [1, 2, 3, 4, 128])
@pytest.mark.parametrize("cutoff", [0.1, 1.0, 3.0, 4.9])
@pytest.mark.parametrize("loop", [True, False])
@pytest.mark.parametrize("include_transpose", [True, False])
@pytest.mark.parametrize("box_type", [None, "triclinic", "rectangular"])
def test_neighbors(hip_lib, strategy, n_batches, cutoff, loop, include_transpose, box_type):
    from oracle.neighbors_numpy import reference_neighbors, sort_neighbors
    from torchmdnet_amd.models.utils import OptimizedDistance

    # (the reference skips cell + triclinic, tests/test_neighbors.py:88: its cell kernel is orthorhombic-only; tn_cell.hip is not)
    lbox = 10.0
    pos, batch = _system(n_batches, lbox=lbox)
    box = None
    if box_type == "rectangular":
        box = torch.tensor([[lbox, 0.0, 0.0], [0.0, lbox, 0.0], [0.0, 0.0, lbox]])
    elif box_type == "triclinic":
        box = torch.tensor([[lbox, 0.0, 0.0], [0.1, lbox, 0.0], [0.3, 0.2, lbox]])
    ref_nb, ref_vec, ref_d = reference_neighbors(pos.numpy(), batch.numpy(), loop, include_transpose, cutoff,
                                                 None if box is None else box.numpy())
    max_num_pairs = max(ref_nb.shape[1], 1)
    nl = OptimizedDistance(cutoff_lower=0.0, loop=loop, cutoff_upper=cutoff, max_num_pairs=max_num_pairs, strategy=strategy,
                           box=box, return_vecs=True, include_transpose=include_transpose, resize_to_fit=False).to("cuda")
    nb, d, vec = nl(pos.cuda(), batch.cuda())
    assert nb.shape == (2, max_num_pairs) and d.shape == (max_num_pairs,) and vec.shape == (max_num_pairs, 3)
    nb, vec, d = nb.cpu().numpy(), vec.cpu().numpy(), d.cpu().numpy()
    valid = nb[0] >= 0
    assert valid.sum() == ref_nb.shape[1]
    nb, vec, d = sort_neighbors(nb[:, valid], vec[valid], d[valid])
    assert np.array_equal(nb, ref_nb)
    assert np.allclose(d, ref_d) and np.allclose(vec, ref_vec)


def test_large_and_per_molecule_boxes_and_lower_cutoff(hip_lib):
    from oracle.neighbors_numpy import reference_neighbors, sort_neighbors
    from torchmdnet_amd.models.utils import OptimizedDistance

    torch.manual_seed(1)
    # 3200 atoms in one molecule (reference tests/test_neighbors.py:320-375)
    pos = torch.rand(3200, 3) * 30.0
    batch = torch.zeros(3200, dtype=torch.long)
    ref = reference_neighbors(pos.numpy(), batch.numpy(), False, True, 2.5, None, cutoff_lower=1.0)
    nl = OptimizedDistance(cutoff_lower=1.0, cutoff_upper=2.5, max_num_pairs=-64, return_vecs=True).to("cuda")
    nb, d, vec = nl(pos.cuda(), batch.cuda())
    got = sort_neighbors(nb.cpu().numpy(), vec.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(got[0], ref[0]) and np.allclose(got[2], ref[2])
    # one box per molecule (reference :608-669)
    pos, batch = _system(3)
    boxes = torch.stack([torch.eye(3) * L for L in (10.0, 12.0, 15.0)])
    ref = reference_neighbors(pos.numpy(), batch.numpy(), True, True, 4.0, boxes.numpy())
    nl = OptimizedDistance(cutoff_upper=4.0, max_num_pairs=-128, return_vecs=True, loop=True).to("cuda")
    nb, d, vec = nl(pos.cuda(), batch.cuda(), box=boxes.cuda())
    got = sort_neighbors(nb.cpu().numpy(), vec.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(got[0], ref[0]) and np.allclose(got[1], ref[1]) and np.allclose(got[2], ref[2])


def test_pair_overflow_raises_runtime_error(hip_lib):
    """reference tests/test_model_utils.py:71-88"""
    from torchmdnet_amd.models.utils import OptimizedDistance

    pos = torch.zeros(50, 3, device="cuda") + torch.rand(50, 3, device="cuda") * 0.1
    nl = OptimizedDistance(cutoff_upper=5.0, max_num_pairs=10)
    with pytest.raises(RuntimeError, match="max_num_pairs"):
        nl(pos)


def test_neighbor_gradients(hip_lib):  # backward = tmdnet::neighbor_grad (HIP kernel)
    """d(deltas, distances)/d(pos) vs a pure-torch evaluation on the same pair list
    (reference tests/test_neighbors.py:151-261; fp32 tolerance 1e-2 there)."""
    from torchmdnet_amd.models.utils import OptimizedDistance

    torch.manual_seed(3)
    pos = (torch.rand(200, 3) * 8.0).cuda().requires_grad_(True)
    batch = torch.zeros(200, dtype=torch.long, device="cuda")
    nl = OptimizedDistance(cutoff_upper=3.0, max_num_pairs=-64, return_vecs=True, loop=True).to("cuda")
    nb, d, vec = nl(pos, batch)
    (d.sum() + (vec * vec).sum()).backward()
    got = pos.grad.clone()
    p2 = pos.detach().clone().requires_grad_(True)
    v2 = p2[nb[0]] - p2[nb[1]]
    nz = nb[0] != nb[1]
    d2 = v2[nz].norm(dim=-1)
    (d2.sum() + (v2 * v2).sum()).backward()
    assert torch.allclose(got, p2.grad, atol=1e-4, rtol=1e-4)
