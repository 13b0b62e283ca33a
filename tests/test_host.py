"""not-gpu: host-side logic of the drop-in surface (no compute calls: there is no CPU path)."""
import copy
import ctypes as C
import os

import pytest
import torch

from oracle import ref_shims as R
from torchmdnet_amd import workloads as W
from torchmdnet_amd.models.model import Ensemble, TorchMD_Net, create_model, load_model


def test_library_loads_and_exports_every_declared_symbol(hip_lib):
    from torchmdnet_amd import _C

    syms = _C.check_symbols()
    assert len(syms) >= 20 and "tmdnet_energy_forces" in syms and "tmdnet_build_graph" in syms
    assert hip_lib.tmdnet_version().decode().startswith("tmdnet_amd")
    # parameter table of the C side == state-dict keys of the Python side
    model = create_model(dict(W.TINY_ARGS))
    hp = model._hparams()
    handle = C.c_void_p()
    assert hip_lib.tmdnet_create(C.byref(hp), C.byref(handle)) == 0
    names = set()
    sd = model.state_dict()
    for i in range(hip_lib.tmdnet_num_params(handle)):
        numel = C.c_int64()
        name = hip_lib.tmdnet_param_name(handle, i, C.byref(numel)).decode()
        assert name in sd and sd[name].numel() == numel.value, name
        names.add(name)
    skipped = set(sd) - names
    assert skipped == {"representation_model.distance.box"}, skipped  # geometry input, not a weight
    # wrong size / unknown name are rejected with a message
    buf = (C.c_float * 3)()
    assert hip_lib.tmdnet_set_param(handle, b"mean", buf, 3) != 0
    assert b"expected 1" in hip_lib.tmdnet_last_error(handle)
    assert hip_lib.tmdnet_set_param(handle, b"nope", buf, 3) != 0
    assert hip_lib.tmdnet_finalize_params(handle) != 0  # parameters missing -> state error, no crash
    hip_lib.tmdnet_destroy(handle)


def test_seeded_init_matches_reference_checksums(golden_dir):
    """torch.manual_seed(s); create_model(args) gives the reference's weights (tests/test_model.py:265-274)."""
    g = torch.load(os.path.join(golden_dir, "c2_ref.pt"))
    torch.manual_seed(0)
    m = create_model(dict(W.C2_ARGS))
    chk = float(sum(v.double().abs().sum() for v in m.state_dict().values() if v.is_floating_point()))
    assert abs(chk - g["sd_checksum"]) < 1e-6 * g["sd_checksum"]
    assert sum(p.numel() for p in m.parameters()) == 766337
    g2 = torch.load(os.path.join(golden_dir, "expected_tensornet_scalar.pt"))
    R.seed_everything(1234)
    m2 = create_model(dict(g2["args"]))
    chk2 = float(sum(v.double().abs().sum() for v in m2.state_dict().values() if v.is_floating_point()))
    assert abs(chk2 - g2["sd_checksum"]) < 1e-6 * g2["sd_checksum"]
    # Equivariant Transformer: ET-SPICE hyper-parameters (seed 0) and the reference's golden-vector recipe (seed 1234)
    g3 = torch.load(os.path.join(golden_dir, "et_c4_ref.pt"))
    torch.manual_seed(0)
    m3 = create_model(dict(W.C4_ARGS))
    chk3 = float(sum(v.double().abs().sum() for v in m3.state_dict().values() if v.is_floating_point()))
    assert abs(chk3 - g3["sd_checksum"]) < 1e-6 * g3["sd_checksum"]
    g4 = torch.load(os.path.join(golden_dir, "expected_et_scalar.pt"))
    R.seed_everything(1234)
    m4 = create_model(dict(g4["args"]))
    chk4 = float(sum(v.double().abs().sum() for v in m4.state_dict().values() if v.is_floating_point()))
    assert abs(chk4 - g4["sd_checksum"]) < 1e-6 * g4["sd_checksum"]


@pytest.mark.skipif(not R.reference_available(), reason="/root/reference not present")
def test_state_dict_identical_to_live_reference():
    mm = R.reference_model_module()
    for args, seed in [(W.TINY_ARGS, 3), (dict(W.TINY_ARGS, prior_model="Atomref", prior_args={"max_z": 20}), 4),
                       (W.ET_TINY_ARGS, 5), (dict(W.ET_TINY_ARGS, neighbor_embedding=False, distance_influence="values"), 6)]:
        torch.manual_seed(seed)
        a = create_model(dict(args))
        torch.manual_seed(seed)
        b = mm.create_model(dict(args))
        sa, sb = a.state_dict(), b.state_dict()
        assert set(sa) == set(sb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


def _write_ckpt(path, model, args, old_layout=False):
    sd = {"model." + k: v.clone() for k, v in model.state_dict().items()}
    hp = dict(args)
    if old_layout:
        hp["check_errors"] = True
        F = args["embedding_dimension"]
        keys = ["model.representation_model.tensor_embedding.linears_scalar.1"]
        keys += [f"model.representation_model.layers.{l}.linears_scalar.2" for l in range(args["num_layers"])]
        for k in keys:  # current row c*F+f  ->  old row f*3+c  (inverse of the reference's remix, model.py:321-331)
            w, b = sd[k + ".weight"], sd[k + ".bias"]
            sd[k + ".weight"] = w.view(3, F, -1).transpose(0, 1).reshape(3 * F, -1).contiguous()
            sd[k + ".bias"] = b.view(3, F).transpose(0, 1).reshape(3 * F).contiguous()
        # legacy head key names and missing box buffer (model.py:261-294)
        for old, new in [("model.output_model.output_network.0.", "model.output_model.output_network.layers.0."),
                         ("model.output_model.output_network.2.", "model.output_model.output_network.layers.2.")]:
            for suffix in ("weight", "bias"):
                sd[old + suffix] = sd.pop(new + suffix)
        sd.pop("model.representation_model.distance.box")
    torch.save({"state_dict": sd, "hyper_parameters": hp}, path)


@pytest.mark.parametrize("old_layout", [False, True])
def test_load_model_checkpoint_formats(tmp_path, old_layout):
    torch.manual_seed(5)
    args = dict(W.TINY_ARGS)
    model = create_model(dict(args))
    path = str(tmp_path / "m.ckpt")
    _write_ckpt(path, model, args, old_layout=old_layout)
    with pytest.warns(UserWarning) if old_layout else _nullcontext():
        loaded = load_model(path, derivative=True)
    for k, v in model.state_dict().items():
        assert torch.equal(loaded.state_dict()[k], v), k
    assert loaded.derivative is True
    if R.reference_available():  # the reference reads the same file to the same tensors
        mm = R.reference_model_module()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = mm.load_model(path, derivative=True)
        for k, v in ref.state_dict().items():
            assert torch.equal(loaded.state_dict()[k], v), k


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_load_ensemble_and_atomref_toggle(tmp_path):
    args = dict(W.TINY_ARGS, prior_model="Atomref", prior_args={"max_z": 20, "enable": False}, remove_ref_energy=True)
    paths = []
    for s in (1, 2):
        torch.manual_seed(s)
        m = create_model(dict(args))
        p = str(tmp_path / f"m{s}.ckpt")
        _write_ckpt(p, m, args)
        paths.append(p)
    ens = load_model(paths, return_std=True)
    assert isinstance(ens, Ensemble) and len(ens) == 2 and ens.return_std
    one = load_model(paths[0])
    assert one.prior_model[-1].enable is False and one._atomref_table() is None
    one = load_model(paths[0], remove_ref_energy=False)  # total energies: Atomref switched on (model.py:250-259)
    assert one.prior_model[-1].enable is True and one._atomref_table().shape == (20,)


def test_no_cpu_fallback_and_argument_checks():
    model = create_model(dict(W.TINY_ARGS))
    z, pos, batch = W.synthetic_batch(n_mol=1, n_atoms=5)
    with pytest.raises(RuntimeError, match="no CPU"):
        model(z, pos, batch)
    with pytest.raises(AssertionError):
        model(z.to(torch.int32), pos, batch)
    with pytest.raises(ValueError, match="Unknown architecture"):
        create_model(dict(W.TINY_ARGS, model="nonsense"))
    with pytest.raises(NotImplementedError):
        create_model(dict(W.TINY_ARGS, model="graph-network"))
    with pytest.raises(NotImplementedError):
        create_model(dict(W.TINY_ARGS, precision=64))


def test_module_protocol_deepcopy_and_freezing():
    model = create_model(dict(W.TINY_ARGS))
    assert isinstance(model, TorchMD_Net)
    clone = copy.deepcopy(model)  # tests/test_calculator.py:54 of the reference does this
    assert clone._engine is not model._engine and clone._engine.handle is None
    for p in clone.parameters():
        p.requires_grad = False  # calculators.py:226-227
    clone.eval()
    assert clone.representation_model.static_shapes is False and clone.representation_model.distance.resize_to_fit is True
    assert set(clone.state_dict()) == set(model.state_dict())


def test_install_as_torchmdnet_alias():
    import sys

    import torchmdnet_amd

    saved = {k: v for k, v in sys.modules.items() if k == "torchmdnet" or k.startswith("torchmdnet.")}
    saved_installed = R._installed
    try:
        for k in saved:
            del sys.modules[k]
        torchmdnet_amd.install_as_torchmdnet()
        from torchmdnet.models.model import create_model as cm  # noqa: F401

        assert cm is create_model
        # the reference's adapters (TorchMD `External`, ASE `TMDNETCalculator`) are loaded from the reference's own file
        # and bind to this package's load_model; `ase` is stubbed when it is not installed (it only provides a base class)
        if R.reference_available():
            import types

            stubbed = []
            try:
                import ase.calculators.calculator  # noqa: F401
            except ImportError:
                for name in ("ase", "ase.calculators", "ase.calculators.calculator"):
                    sys.modules[name] = types.ModuleType(name)
                    stubbed.append(name)
                sys.modules["ase.calculators.calculator"].Calculator = object
                sys.modules["ase.calculators.calculator"].all_changes = []
            try:
                for k in [k for k in sys.modules if k == "torchmdnet" or k.startswith("torchmdnet.")]:
                    del sys.modules[k]
                done = torchmdnet_amd.install_as_torchmdnet(reference_root=R.REFERENCE_ROOT)
                assert "torchmdnet.calculators" in done
                calc = sys.modules["torchmdnet.calculators"]
                from torchmdnet_amd.models.model import load_model as lm

                assert calc.load_model is lm and hasattr(calc, "External") and hasattr(calc, "TMDNETCalculator")
                # External with a module instead of a path: the adapter's own code runs up to the first model call
                ext = calc.External(create_model(dict(W.TINY_ARGS)), torch.ones(1, 5, dtype=torch.long), device="cpu")
                assert ext.n_atoms == 5 and ext.model.training is False
            finally:
                for name in stubbed:
                    sys.modules.pop(name, None)
    finally:
        for k in [k for k in sys.modules if k == "torchmdnet" or k.startswith("torchmdnet.")]:
            del sys.modules[k]
        sys.modules.update(saved)
        R._installed = saved_installed and "torchmdnet.extensions.ops" in saved
        assert R.purge_foreign_torchmdnet() == 0, "the torchmdnet alias leaked out of this test"


def test_load_model_equivariant_transformer_checkpoint(tmp_path):
    """An Equivariant Transformer Lightning checkpoint as the reference writes it (hyper_parameters + "model."-prefixed state
    dict) loads to the same tensors here and in the live reference."""
    torch.manual_seed(9)
    args = dict(W.ET_TINY_ARGS)
    model = create_model(dict(args))
    path = str(tmp_path / "et.ckpt")
    # 'check_errors' was a global training argument: old ET checkpoints carry it too, and must NOT get the TensorNet remix
    # (reference model.py:333-372 applies it to tensornet / tensornet2 only)
    torch.save({"state_dict": {"model." + k: v.clone() for k, v in model.state_dict().items()},
                "hyper_parameters": dict(args, check_errors=True)}, path)
    with pytest.warns(UserWarning, match="Old-format"):
        loaded = load_model(path, derivative=True)
    assert type(loaded.representation_model).__name__ == "TorchMD_ET"
    for k, v in model.state_dict().items():
        assert torch.equal(loaded.state_dict()[k], v), k
    if R.reference_available():
        mm = R.reference_model_module()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = mm.load_model(path, derivative=True)
        assert set(ref.state_dict()) == set(loaded.state_dict())
        for k, v in ref.state_dict().items():
            assert torch.equal(loaded.state_dict()[k], v), k


def test_fingerprint_cache_follows_module_changes():
    """The cached tensor list behind the parameter-change detector is dropped by .to()/.float() (new buffer objects) and
    sees in-place updates (load_state_dict, optimiser steps) through the tensors' version counters."""
    model = create_model(dict(W.TINY_ARGS))
    fp0 = model._fingerprint()
    assert model._fingerprint() == fp0
    with torch.no_grad():
        model.mean.add_(1.0)
    fp1 = model._fingerprint()
    assert fp1 != fp0
    model.load_state_dict(copy.deepcopy(model.state_dict()))
    assert model._fingerprint() != fp1
    model.double()
    assert model._engine.tensors is None  # rebuilt lazily with the new tensor objects
    assert len(model._fingerprint()) == len(fp0)
    # replaced tensor OBJECTS: same version / possibly same address, different identity
    model.float()
    fp2 = model._fingerprint()
    model.load_state_dict(copy.deepcopy(model.state_dict()), assign=True)
    fp3 = model._fingerprint()
    assert fp3 != fp2
    lin = model.representation_model.linear
    lin.weight = torch.nn.Parameter(lin.weight.detach().clone())
    fp4 = model._fingerprint()
    assert fp4 != fp3
    model.output_model = copy.deepcopy(model.output_model)  # swapped submodule
    assert model._fingerprint() != fp4
    assert model._fingerprint() == model._fingerprint()
    assert "equivariant-transformer" in __import__("torchmdnet_amd.models", fromlist=["x"]).__all_models__


def test_custom_ops_are_registered_with_fake_impls():
    """torch.ops.tmdnet.{energy_forces, neighbor_pairs, neighbor_grad} exist with fake (meta) implementations: shape
    propagation works on fake CUDA tensors without a GPU - what torch.compile / torch.export need to trace the model
    (reference: register_fake of its Warp ops, warp_ops/neighbors.py:61-104)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    from torchmdnet_amd import ops  # noqa: F401

    with FakeTensorMode():
        z = torch.empty(7, dtype=torch.long, device="cuda")
        pos = torch.empty(7, 3, device="cuda")
        b = torch.zeros(7, dtype=torch.long, device="cuda")
        e, f = torch.ops.tmdnet.energy_forces(z, pos, b, None, None, 1, 2, True)
        assert e.shape == (2,) and f.shape == (7, 3) and e.device.type == "cuda"
        e, f = torch.ops.tmdnet.energy_forces(z, pos, b, None, None, 1, 2, False)
        assert f.shape == (0, 3)
        nb, d, w, n = torch.ops.tmdnet.neighbor_pairs(pos, b, None, 0.0, 5.0, 40, True, True, 0, 1)
        assert nb.shape == (2, 40) and nb.dtype == torch.long and d.shape == (40, 3) and w.shape == (40,) and n.dtype == torch.int32
        g = torch.ops.tmdnet.neighbor_grad(nb, d, w, d, w, 7)
        assert g.shape == (7, 3)
    model = create_model(dict(W.TINY_ARGS))
    clone = copy.deepcopy(model)
    assert clone._engine.op_key is None and model._engine.op_key is None  # keys are handed out on first use, never shared


def test_tensornet2_host_mirror_matches_reference_init_and_keys(hip_lib):
    """TensorNet2 + ScalarPlusWeightedCoulomb (AceFF-2.0): same seed -> same state dict as the unmodified reference (keys, shapes,
    values), with and without a Coulomb cutoff; the C side expects exactly those tensors."""
    args = dict(W.TINY_ARGS, model="tensornet2", output_model="ScalarPlusWeightedCoulomb", q_dim=8, q_weights=[1.0, 0.5, 2.0])
    for extra in ({}, {"coulomb_cutoff": 6.0}):
        a = dict(args, **extra)
        torch.manual_seed(17)
        mine = create_model(dict(a))
        sd = mine.state_dict()
        if R.reference_available():
            mm = R.reference_model_module()
            torch.manual_seed(17)
            ref = mm.create_model(dict(a)).state_dict()
            assert set(ref) == set(sd), set(ref) ^ set(sd)
            for k, v in ref.items():
                assert torch.equal(v, sd[k]), k
        hp = mine._tn2_hparams()
        handle = C.c_void_p()
        assert hip_lib.tmdnet_create_tn2(C.byref(hp), C.byref(handle)) == 0
        names = set()
        for i in range(hip_lib.tmdnet_num_params(handle)):
            numel = C.c_int64()
            name = hip_lib.tmdnet_param_name(handle, i, C.byref(numel)).decode()
            assert name in sd and sd[name].numel() == numel.value, name
            names.add(name)
        assert set(sd) - names <= {"representation_model.distance.box", "output_model.distance.box"}
        hip_lib.tmdnet_destroy(handle)
    with pytest.raises(NotImplementedError):
        create_model(dict(args, output_model="Scalar"))


def test_parameter_gradient_entries_map_back_to_state_dict_rows():
    """torchmdnet_amd/models/model.py::_et_grads undoes the engine's packing (csrc/tn_et_api.hip `thirds`: value-type rows from the
    reference's per-head [H][3][hd] order to thirds [3][F]); with entries that carry their own engine row index every state-dict
    row must come back in place.  No GPU: the entries are made up here."""
    import torch
    from torchmdnet_amd import workloads as W
    from torchmdnet_amd.models.model import create_model

    args = dict(W.ET_TINY_ARGS)
    model = create_model(dict(args))
    rm = model.representation_model
    F, hd, K = rm.hidden_channels, rm.hidden_channels // rm.num_heads, args["num_rbf"]
    Z, F2 = rm.embedding.weight.shape[0], F // 2

    def thirds_src(dst_row):  # state-dict row of engine row t * F + c (the C++ packer, restated)
        t, c = divmod(dst_row, F)
        return (c // hd) * 3 * hd + t * hd + c % hd

    ent = {"emb": torch.zeros(Z * F), "embN": torch.zeros(Z * F), "Wn": torch.zeros(F * K), "bn": torch.zeros(F), "Wc": torch.zeros(2 * F * F),
           "bc": torch.zeros(F), "lno_w": torch.zeros(F), "lno_b": torch.zeros(F), "W1u": torch.zeros((F + F2) * F), "Wm1": torch.zeros(2 * F * F),
           "bm1": torch.zeros(F), "Wm2": torch.zeros(F * F), "bm2": torch.zeros(F), "W21": torch.zeros(F2 * F2), "Wn1": torch.zeros(F2 * F),
           "bn1": torch.zeros(F2), "Wn2": torch.arange(F2, dtype=torch.float32), "bn2": torch.tensor([7.0])}
    for l in range(args["num_layers"]):
        t = "l%d." % l
        ent.update({t + "ln_w": torch.zeros(F), t + "ln_b": torch.zeros(F), t + "Wvp": torch.zeros(3 * F * F), t + "Wo": torch.zeros(3 * F * F),
                    t + "bo": torch.zeros(3 * F)})
        ent[t + "bqkv"] = torch.arange(5 * F, dtype=torch.float32)                      # value = engine row
        ent[t + "Wqkv"] = torch.arange(5 * F, dtype=torch.float32).repeat_interleave(F)
        ent[t + "bdkv"] = torch.arange(4 * F, dtype=torch.float32)
        ent[t + "Wdkv"] = torch.arange(4 * F, dtype=torch.float32).repeat_interleave(K)
    grads = model._et_grads(ent)
    al = rm.attention_layers[0]
    assert torch.equal(grads[al.q_proj.bias], torch.arange(F, dtype=torch.float32))
    assert torch.equal(grads[al.k_proj.bias], torch.arange(F, 2 * F, dtype=torch.float32))
    gv, gdv = grads[al.v_proj.bias], grads[al.dv_proj.bias]
    for dst in range(3 * F):
        assert gv[thirds_src(dst)] == 2 * F + dst and gdv[thirds_src(dst)] == F + dst
    assert torch.equal(grads[al.v_proj.weight][:, 0], gv) and torch.equal(grads[al.dk_proj.bias], torch.arange(F, dtype=torch.float32))
    last = model.output_model.output_network[1].update_net.layers[2]
    assert torch.equal(grads[last.weight][0], torch.arange(F2, dtype=torch.float32)) and grads[last.weight][1].abs().max() == 0
    assert grads[last.bias][0] == 7.0 and grads[last.bias][1] == 0.0
    assert set(id(p) for p in grads) <= set(id(p) for p in model.parameters())


def test_every_developer_switch_is_listed_for_the_gpu_switch_tests():
    """tests/test_gpu_switches.py runs every TMDNET_* environment switch on its own: a switch added to the library without an
    entry there fails HERE (CPU), so no kernel variant can hide behind an untested switch."""
    import glob, re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for path in glob.glob(os.path.join(root, "torchmd-net_amd", "csrc", "*")) + glob.glob(os.path.join(root, "torchmd-net_amd", "torchmdnet_amd", "**", "*.py"), recursive=True):
        found |= set(re.findall(r'getenv\("(TMDNET_[A-Z0-9_]+)"\)|environ(?:\.get)?\(?\[?"(TMDNET_[A-Z0-9_]+)"', open(path).read())) 
    names = {a or b for a, b in found}
    import importlib.util
    spec = importlib.util.spec_from_file_location("sw", os.path.join(root, "tests", "test_gpu_switches.py"))
    sw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sw)
    missing = names - set(sw.SWITCHES) - sw.NOT_KERNEL_SWITCHES
    assert not missing, f"switches without a GPU test: {sorted(missing)}"
    assert set(sw.SWITCHES) <= names, f"stale entries: {sorted(set(sw.SWITCHES) - names)}"


def test_force_matching_backward_combines_the_passes_with_the_right_signs(monkeypatch):
    """Host logic of _EnergyForceParamGrad.backward with the engine calls replaced by stand-ins: loss = sum g_E . E + sum g_F . F
    -> d/d theta = [energy pass seeded with g_E] - [second-order pass along g_F], d/d pos = - g_E[batch] F - H g_F (reference:
    autograd through model.py:618-628 with create_graph=True); the difference-quotient orders call the first-order pass 2 / 4 more
    times and announce the missing position term once; force_position_gradient = False skips H v and announces it too."""
    import contextlib
    import warnings

    import torchmdnet_amd.models.model as M
    from torchmdnet_amd import workloads as W

    torch.manual_seed(0)
    model = M.create_model(dict(W.TINY_ARGS, derivative=True))
    params = [p for p in model.parameters() if p.requires_grad]
    n, n_mol = 7, 2
    z, batch = torch.ones(n, dtype=torch.long), torch.tensor([0, 0, 0, 1, 1, 1, 1])
    E0, F0 = torch.tensor([1.5, -2.0]), torch.randn(n, 3)
    hv0 = torch.randn(n, 3)
    calls = dict(first=0, second=0, hv=0)

    def energy_and_forces(*a, **k):
        return E0.clone(), F0.clone()

    def parameter_gradients_of(z_, pos_, batch_, box, q, n_mol_, ge):
        calls["first"] += 1
        return E0, {p: torch.full_like(p, float(ge.sum())) for p in params}

    def force_term_parameter_gradients(z_, pos_, batch_, box, q, n_mol_, v, want_hv=False, ge=None, scale=1.0):
        # with an energy seed: the gradients of S = s - sum ge E (parameters: minus the energy pass' stand-in; positions: + ge F)
        calls["second"] += 1
        calls["hv"] += int(want_hv)
        calls["seeded"] = calls.get("seeded", 0) + int(ge is not None)
        g = {p: scale * torch.full_like(p, float(v.sum()) - (0.0 if ge is None else float(ge.sum()))) for p in params}
        hv = scale * (hv0.clone() if ge is None else hv0 + ge[batch_].unsqueeze(1) * F0)  # (the engine scales its flat output once)
        return (g, hv) if want_hv else g

    monkeypatch.setattr(M, "_direct_radial_functions", lambda m: contextlib.nullcontext())
    for name, fn in (("energy_and_forces", energy_and_forces), ("parameter_gradients_of", parameter_gradients_of),
                     ("force_term_parameter_gradients", force_term_parameter_gradients)):
        monkeypatch.setattr(model, name, fn)
    ge, R = torch.tensor([0.5, 2.0]), torch.randn(n, 3)

    def run(order, pos_grad=True, one_pass=False):
        model.force_gradient_order, model.force_position_gradient, model._warned_pos_grad = order, pos_grad, False
        model.one_pass_training = one_pass
        for p in params:
            p.grad = None
        for k in calls:
            calls[k] = 0
        pos = torch.zeros(n, 3, requires_grad=True)
        y, F = M._EnergyForceParamGrad.apply(model, z, pos, batch, None, None, n_mol, *params)
        ((y * ge).sum() + (F * R).sum()).backward()
        return pos.grad

    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the analytic pass with H v has nothing to announce
        g_pos = run(0)
    assert calls == dict(first=1, second=1, hv=1, seeded=0)
    expect = float(ge.sum()) - float(R.sum())  # energy pass seeded with g_E, minus the second-order pass along v = g_F
    assert all(torch.allclose(p.grad, torch.full_like(p, expect), rtol=1e-5, atol=1e-5) for p in params)
    assert torch.allclose(g_pos, -ge[batch].unsqueeze(1) * F0 - hv0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # force_position_gradient = False is the caller's own choice: silent
        g_pos = run(0, pos_grad=False)
    assert calls == dict(first=1, second=1, hv=0, seeded=0) and torch.allclose(g_pos, -ge[batch].unsqueeze(1) * F0)
    # one pass (the default): no first-order pass at all, the seeded second-order pass delivers both terms, same numbers
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        g_pos = run(0, one_pass=True)
    assert calls == dict(first=0, second=1, hv=1, seeded=1)
    assert all(torch.allclose(p.grad, torch.full_like(p, expect), rtol=1e-5, atol=1e-5) for p in params)
    assert torch.allclose(g_pos, -ge[batch].unsqueeze(1) * F0 - hv0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        g_pos = run(0, pos_grad=False, one_pass=True)
    assert calls == dict(first=0, second=1, hv=0, seeded=1) and torch.allclose(g_pos, -ge[batch].unsqueeze(1) * F0)
    assert all(torch.allclose(p.grad, torch.full_like(p, expect), rtol=1e-5, atol=1e-5) for p in params)
    calls.pop("seeded")
    for order, extra in ((2, 2), (4, 4)):  # (None means 0 for every architecture: covered by the first run's explicit 0)
        with pytest.warns(UserWarning, match="energy term's part"):
            g_pos = run(order)
        assert calls == dict(first=1 + extra, second=0, hv=0) and torch.allclose(g_pos, -ge[batch].unsqueeze(1) * F0)
        # the stand-in's first-order gradient does not depend on pos: every difference quotient of it is zero
        assert all(torch.allclose(p.grad, torch.full_like(p, float(ge.sum())), rtol=1e-5, atol=1e-5) for p in params)
