"""-m gpu: the embedding in the radial basis (csrc/tn_embed_rb.hip: species-resolved moments of the radial functions per atom +
two per-atom MFMA contractions instead of per-pair Q / dQ rows) against the per-pair form it replaces (same library, option
"embed_rb_min_atoms" = huge) and against the oracle.  Reference math: TensorEmbedding.forward, tensornet.py:543-619, 405-445,
526-541.  Both forms evaluate the same sums in a different order: they agree to rounding (bounds below), the oracle bound is the
north-star 1e-4."""
import pytest
import torch

from torchmdnet_amd import workloads as W

pytestmark = pytest.mark.gpu
REL = 1e-4


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _pair(args, seed=0):
    """the same weights twice: radial-basis embedding (default) and the per-pair form"""
    from torchmdnet_amd.models.model import create_model

    torch.manual_seed(seed)
    rb = create_model(dict(args)).to("cuda")
    old = create_model(dict(args))
    old.load_state_dict(rb.state_dict())
    old = old.to("cuda")
    old.set_engine_option("embed_rb_min_atoms", 10 ** 12)
    return rb, old


def _species(z, n_species):
    """map the generator's 4 elements onto `n_species` atomic numbers (deterministic)"""
    idx = torch.arange(z.shape[0])
    return ((idx * 7 + idx // 5 + z) % n_species) * 2 + 1


@pytest.mark.parametrize("F,K,n_species", [(128, 32, 4), (128, 32, 3), (64, 32, 7), (64, 64, 4), (128, 64, 8), (128, 32, 1)])
def test_radial_basis_embedding_equals_pair_form(hip_lib, F, K, n_species):
    args = dict(W.C2_ARGS, embedding_dimension=F, num_rbf=K, max_z=40)
    rb, old = _pair(args, seed=F + K)
    z, pos, batch = W.synthetic_batch(n_mol=40, n_atoms=40, first_seed=300)  # 1600 atoms >= 1024
    z = _species(z, n_species)
    zc, pc, bc = z.cuda(), pos.cuda(), batch.cuda()
    E, Fo = rb(zc, pc, bc)
    assert rb.engine_info("embed_rb") == 1.0 and rb.engine_info("species_last_build") == n_species
    n = z.shape[0]
    u0 = rb.debug_tensor("u0", (n, 9, F))
    X0 = rb.debug_tensor("X_embed", (n, 9, F))
    Eo, Fold = old(zc, pc.clone(), bc)
    assert old.engine_info("species_last_build") == 0  # the per-pair form ran
    assert rel_err(u0, old.debug_tensor("u0", (n, 9, F))) < 5e-6
    assert rel_err(X0, old.debug_tensor("X_embed", (n, 9, F))) < 5e-6
    assert rel_err(E, Eo) < 5e-6 and rel_err(Fo, Fold) < 2e-5, (rel_err(E, Eo), rel_err(Fo, Fold))
    E2, F2 = rb(zc, pc.clone(), bc)
    assert torch.equal(E, E2) and torch.equal(Fo, F2)  # deterministic
    # energies only (no reverse pass, no gradient buffers)
    rb.derivative = False
    with torch.no_grad():
        y, _ = rb(zc, pc.detach().clone(), bc)
    assert rel_err(y, Eo) < 5e-6


def test_radial_basis_embedding_vs_oracle_charges_ragged(hip_lib):
    """ragged molecules (1 .. 90 atoms), total charges, 5 species: oracle on every molecule"""
    from oracle import tensornet_torch as T
    from torchmdnet_amd.models.model import create_model

    args = dict(W.C2_ARGS, max_z=40)
    torch.manual_seed(3)
    model = create_model(dict(args)).to("cuda")
    sizes = [90, 1, 33, 64, 2, 17] * 6  # 1242 atoms
    zs, ps, bs = [], [], []
    for m, n in enumerate(sizes):
        zz, pp = W.synthetic_molecule(7000 + m, n_atoms=n)
        zs.append(torch.from_numpy(zz))
        ps.append(torch.from_numpy(pp))
        bs.append(torch.full((n,), m, dtype=torch.long))
    z, pos, batch = _species(torch.cat(zs), 5), torch.cat(ps), torch.cat(bs)
    q = torch.tensor([float(m % 3 - 1) for m in range(len(sizes))])
    E, F = model(z.cuda(), pos.cuda(), batch.cuda(), q=q.cuda())
    assert model.engine_info("species_last_build") == 5
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    Er, Fr = T.energy_and_forces(sd, T.hparams_from_args(args), z, pos, batch, q=q)
    assert rel_err(E.cpu(), Er) < REL and rel_err(F.cpu(), Fr) < REL
    net = torch.zeros(len(sizes), 3).index_add(0, batch, F.cpu())
    assert net.abs().max().item() < 1e-3 * F.abs().max().item()


def test_radial_basis_embedding_periodic_cell_list_and_fallbacks(hip_lib):
    """one periodic system through the cell list (atoms renumbered: the species index follows the internal order); more than
    8 species, static shapes and small systems keep the per-pair form and give the same numbers as before"""
    from oracle import tensornet_c as CO, tensornet_torch as T

    args = dict(W.C2_ARGS, embedding_dimension=64, max_z=40, max_num_neighbors=96)
    rb, old = _pair(args, seed=9)
    z, pos, box = W.water_box(n_side=8, spacing=3.1)  # 1536 atoms
    z = _species(z, 6)
    batch = torch.zeros_like(z)
    zc, pc, bc, xc = z.cuda(), pos.cuda(), batch.cuda(), box.cuda()
    E, F = rb(zc, pc, bc, box=xc)
    assert rb.cell_grid(z.shape[0])[3] == 1 and rb.engine_info("species_last_build") == 6
    Eo, Fo = old(zc, pc.clone(), bc, box=xc)
    assert rel_err(E, Eo) < 5e-6 and rel_err(F, Fo) < 2e-5
    sd = {k: v.detach().cpu() for k, v in rb.state_dict().items()}
    Er, Fr = CO.energy_forces(sd, T.hparams_from_args(args), z, pos, batch, box=box)
    assert rel_err(E.cpu(), Er) < REL and rel_err(F.cpu(), Fr) < REL
    # 11 species: the per-pair tables run, bit-identical to the model that never takes the radial-basis form
    z11 = _species(z, 11).cuda()
    E1, F1 = rb(z11, pc.clone(), bc, box=xc)
    assert rb.engine_info("species_last_build") == 11
    E2, F2 = old(z11, pc.clone(), bc, box=xc)
    assert torch.equal(E1, E2) and torch.equal(F1, F2)
    # a small system stays on the per-pair form as well
    sel = torch.arange(300).cuda()
    E3, F3 = rb(zc[sel], pc[sel].clone(), bc[sel])
    E4, F4 = old(zc[sel], pc[sel].clone(), bc[sel])
    assert rb.engine_info("species_last_build") == 0 and torch.equal(E3, E4) and torch.equal(F3, F4)


def test_radial_basis_embedding_tensornet2(hip_lib):
    """TensorNet2 shares TensorNet's embedding (tensornet2.py:228-463 -> TensorEmbedding): same switch, same agreement; the
    Coulomb head and the charge heads sit on top of it unchanged"""
    args = dict(W.C2_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", embedding_dimension=64, q_dim=8,
                q_weights=[1.0, 0.5, 2.0], max_z=40)
    rb, old = _pair(args, seed=21)
    z, pos, batch = W.synthetic_batch(n_mol=40, n_atoms=40, first_seed=900)
    z = _species(z, 5)
    q = torch.tensor([float(m % 3 - 1) for m in range(40)])
    zc, pc, bc, qc = z.cuda(), pos.cuda(), batch.cuda(), q.cuda()
    E, F = rb(zc, pc, bc, q=qc)
    assert rb.engine_info("species_last_build") == 5
    Eo, Fo = old(zc, pc.clone(), bc, q=qc)
    assert old.engine_info("species_last_build") == 0
    n = z.shape[0]
    assert rel_err(rb.debug_tensor("X_embed", (n, 9, 64)), old.debug_tensor("X_embed", (n, 9, 64))) < 5e-6
    assert rel_err(E, Eo) < 5e-6 and rel_err(F, Fo) < 2e-5, (rel_err(E, Eo), rel_err(F, Fo))
