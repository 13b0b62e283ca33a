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


# ------------------------------------------------------------------------------------------------ float64 operator
# The reference's neighbour kernels are instantiated for float32 AND float64 (neighbors_brute.py:27, tests/test_neighbors.py:83,
# 157, 281): tmdnet_neighbor_pairs_f64 / tmdnet_neighbor_grad_f64 (csrc/tn_neighbors_f64.hip).
@pytest.mark.parametrize("strategy", ["brute", "cell"])
@pytest.mark.parametrize("n_batches", [1, 3, 128])
@pytest.mark.parametrize("cutoff", [0.1, 1.0, 4.9])
@pytest.mark.parametrize("loop,include_transpose", [(True, True), (False, True), (True, False), (False, False)])
@pytest.mark.parametrize("box_type", [None, "triclinic", "rectangular"])
def test_neighbors_float64(hip_lib, strategy, n_batches, cutoff, loop, include_transpose, box_type):
    from oracle.neighbors_numpy import reference_neighbors, sort_neighbors
    from torchmdnet_amd.models.utils import OptimizedDistance

    lbox = 10.0
    pos, batch = _system(n_batches, dtype=torch.float64, lbox=lbox)
    box = None
    if box_type == "rectangular":
        box = torch.tensor([[lbox, 0.0, 0.0], [0.0, lbox, 0.0], [0.0, 0.0, lbox]], dtype=torch.float64)
    elif box_type == "triclinic":
        box = torch.tensor([[lbox, 0.0, 0.0], [0.1, lbox, 0.0], [0.3, 0.2, lbox]], dtype=torch.float64)
    ref_nb, ref_vec, ref_d = reference_neighbors(pos.numpy(), batch.numpy(), loop, include_transpose, cutoff,
                                                 None if box is None else box.numpy())
    max_num_pairs = max(ref_nb.shape[1], 1) + 7  # padding entries behind the list
    nl = OptimizedDistance(cutoff_lower=0.0, loop=loop, cutoff_upper=cutoff, max_num_pairs=max_num_pairs, strategy=strategy,
                           box=box, return_vecs=True, include_transpose=include_transpose, resize_to_fit=False).to("cuda")
    nb, d, vec = nl(pos.cuda(), batch.cuda())
    assert d.dtype == torch.float64 and vec.dtype == torch.float64 and nb.dtype == torch.long
    nb, vec, d = nb.cpu().numpy(), vec.cpu().numpy(), d.cpu().numpy()
    valid = nb[0] >= 0
    assert valid.sum() == ref_nb.shape[1]
    assert (nb[:, ~valid] == -1).all() and (d[~valid] == 0).all() and (vec[~valid] == 0).all()
    nb, vec, d = sort_neighbors(nb[:, valid], vec[valid], d[valid])
    assert np.array_equal(nb, ref_nb)
    assert np.allclose(d, ref_d, rtol=1e-13, atol=1e-13) and np.allclose(vec, ref_vec, rtol=1e-13, atol=1e-13)
    # bit-identical repeats (no atomics in the list build)
    a, b = nl(pos.cuda(), batch.cuda()), nl(pos.cuda(), batch.cuda())
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_float64_per_molecule_boxes_lower_cutoff_and_overflow(hip_lib):
    from oracle.neighbors_numpy import reference_neighbors, sort_neighbors
    from torchmdnet_amd.models.utils import OptimizedDistance

    pos, batch = _system(3, dtype=torch.float64)
    boxes = torch.stack([torch.eye(3, dtype=torch.float64) * L for L in (10.0, 12.0, 15.0)])
    ref = reference_neighbors(pos.numpy(), batch.numpy(), True, True, 4.0, boxes.numpy(), cutoff_lower=0.5)
    nl = OptimizedDistance(cutoff_lower=0.5, cutoff_upper=4.0, max_num_pairs=-128, return_vecs=True, loop=True).to("cuda")
    nb, d, vec = nl(pos.cuda(), batch.cuda(), box=boxes.cuda())
    got = sort_neighbors(nb.cpu().numpy(), vec.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(got[0], ref[0]) and np.allclose(got[1], ref[1], atol=1e-13) and np.allclose(got[2], ref[2], atol=1e-13)
    with pytest.raises(RuntimeError, match="max_num_pairs"):
        OptimizedDistance(cutoff_upper=5.0, max_num_pairs=10)(torch.rand(50, 3, device="cuda", dtype=torch.float64) * 0.1)
    with pytest.raises(RuntimeError, match="fp32 and fp64"):
        OptimizedDistance(cutoff_upper=5.0)(torch.rand(5, 3, device="cuda", dtype=torch.float16))


@pytest.mark.parametrize("loop,include_transpose", [(True, True), (False, False)])
@pytest.mark.parametrize("box_type", [None, "rectangular"])
def test_float64_first_and_second_derivatives(hip_lib, loop, include_transpose, box_type):
    """reference tests/test_neighbors.py:272-315: gradcheck and gradgradcheck through the operator in double precision."""
    from torchmdnet_amd.models.utils import OptimizedDistance

    torch.manual_seed(1234)
    lbox = 10.0
    box = None if box_type is None else (torch.eye(3, dtype=torch.float64) * lbox).cuda()
    nl = OptimizedDistance(cutoff_upper=4.999999, max_num_pairs=100, loop=loop, include_transpose=include_transpose, return_vecs=True,
                           resize_to_fit=True, box=box).to("cuda")
    pos = (0.25 * lbox * torch.rand(10, 3, dtype=torch.float64)).cuda().requires_grad_(True)
    batch = torch.zeros(10, dtype=torch.long, device="cuda")
    f_dist = lambda x: nl(x, batch)[1].sum()  # noqa: E731
    f_both = lambda x: (nl(x, batch)[1] ** 2).sum() + (nl(x, batch)[2] ** 3).sum()  # noqa: E731
    for f in (f_dist, f_both):
        assert torch.autograd.gradcheck(f, (pos,), eps=1e-4, atol=1e-4, rtol=1e-4, nondet_tol=1e-4)
        assert torch.autograd.gradgradcheck(f, (pos,), eps=1e-5, atol=1e-4, rtol=1e-4, nondet_tol=1e-3)
