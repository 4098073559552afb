"""CPU checks of the drop-in boundary: the C-ABI library loads, exports exactly what
include/irbfn_hip.h declares, and rejects bad arguments without touching a GPU; plus the host-side
mirror of the reference interface (model card handling, basis tokens, shape inference)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from irbfn_amd import _lib, configs, flax_rbf
from irbfn_amd.model import WCRBFNet
from irbfn_amd import dynamics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "irbfn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(irbfn_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/irbfn_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header diverge"
    assert lib.irbfn_abi_version() == 1


def test_status_strings_and_arg_validation_without_gpu():
    lib = _lib.load()
    assert lib.irbfn_strerror(0) == b"ok"
    assert b"bad argument" in lib.irbfn_strerror(-1)
    assert lib.irbfn_rollout_state_dim(_lib.ROLLOUT_ST_SELECT) == 7
    assert lib.irbfn_rollout_state_dim(_lib.ROLLOUT_FULLINT) == 5
    assert lib.irbfn_rollout_state_dim(_lib.ROLLOUT_FRENET_LS) == 8
    assert lib.irbfn_rollout_state_dim(_lib.ROLLOUT_SPIRAL) == 6
    assert lib.irbfn_rollout_state_dim(99) == -1
    assert lib.irbfn_rollout_input_dim(_lib.ROLLOUT_ST_KS, 50) == 107
    assert lib.irbfn_rollout_input_dim(_lib.ROLLOUT_FULLINT, 5) == 11
    assert lib.irbfn_rollout_input_dim(_lib.ROLLOUT_SPIRAL, 9) == 5
    # NULL / negative arguments are rejected before any HIP call
    assert lib.irbfn_net_create(None, 7, 1, 10, 2, 0, 0, 0, None, None, None, None, 0) == -1
    h = C.c_void_p()
    assert lib.irbfn_net_create(C.byref(h), 0, 1, 10, 2, 0, 0, 0, None, None, None, None, 0) == -1
    assert lib.irbfn_net_create(C.byref(h), 7, 1, 10, 2, 99, 0, 0, None, None, None, None, 0) == -1
    assert lib.irbfn_net_create(C.byref(h), 9, 1, 10, 2, 0, 0, 0, None, None, None, None, 0) == -2   # D > 8
    assert lib.irbfn_net_forward(None, None, None, 4, None) == -1
    assert lib.irbfn_rollout_forward(99, None, None, None, 4, 5, None) == -1
    assert lib.irbfn_rollout_forward(_lib.ROLLOUT_ST_KS, None, None, None, 0, 5, None) == 0      # B = 0 no-op
    assert lib.irbfn_rollout_forward(_lib.ROLLOUT_ST_KS, None, None, None, 4, 5, None) == -1


def test_check_maps_status_to_python_errors():
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(ValueError):
        _lib.check(-2, "x")
    with pytest.raises(_lib.IrbfnError):
        _lib.check(-5, "x")


def test_basis_tokens():
    assert flax_rbf.basis_name(flax_rbf.gaussian) == "gaussian"
    assert flax_rbf.basis_name("inverse_multiquadric") == "inverse_multiquadric"
    assert set(flax_rbf.NAMES) == set(_lib.BASIS_ENUM)
    with pytest.raises(ValueError):
        flax_rbf.basis_name("gaussian_narrow")      # upstream-only, no source in the reference snapshot
    with pytest.raises(RuntimeError):
        flax_rbf.gaussian(np.zeros(3))              # tokens are not host functions


def test_model_card_validation_and_gate_tables():
    card = configs.model_card(2)
    net = WCRBFNet.from_config(card)
    assert net.config()["basis_func"] == "gaussian" and net.num_split_dimensions == 7
    ns, mr, lo, hi, delta, dr, nr = net._gate_tables()
    assert (ns, mr, nr) == (7, 1, 1) and lo.shape == (7, 1) and dr.shape == (1, 7)
    # ragged bounds (dnmpc_128regions style)
    import json
    cfg = json.load(open(os.path.join(ROOT, "tests/golden/ckpt_dnmpc_128regions.json")))
    net = WCRBFNet.from_config(cfg)
    ns, mr, lo, hi, delta, dr, nr = net._gate_tables()
    assert (ns, mr, nr) == (7, 4, 128) and dr.max() == 3
    np.testing.assert_allclose(lo[2], cfg["lower_bounds"][2])
    bad = dict(card, dimension_ranges=[[0, 0, 0, 0, 0, 0, 5]])
    with pytest.raises(IndexError):
        WCRBFNet.from_config(bad)
    with pytest.raises(ValueError):
        WCRBFNet.from_config(dict(card, basis_func="nope"))
    p = net.init(seed=1)
    assert p["params"]["rbf_list"]["centers"].shape == (128, 10, 7)
    assert (p["params"]["rbf_list"]["log_sigs"] == 0).all()
    with pytest.raises(ValueError):
        net._check_shapes(configs.synth_params(2)["params"])


def test_horizon_inference_and_no_cpu_fallback():
    assert dynamics._infer_T(_lib.ROLLOUT_ST_SELECT, 17) == 5
    assert dynamics._infer_T(_lib.ROLLOUT_FRENET_LS, 108) == 50
    with pytest.raises(ValueError):
        dynamics._infer_T(_lib.ROLLOUT_ST_KS, 16)
    import torch
    if not torch.cuda.is_available():
        net = WCRBFNet.from_config(configs.model_card(1))
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            net.apply(configs.synth_params(1), configs.synth_queries(1, B=4))
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            dynamics.integrate_st_mult(np.zeros((2, 17), np.float32), configs.DYN_PARAMS)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "irbfn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "irbfn_oracle" not in txt.replace("oracle/", ""), f


def test_synthetic_configs_are_deterministic():
    a, b = configs.synth_params(2), configs.synth_params(2)
    assert (a["params"]["linear"]["kernel"] == b["params"]["linear"]["kernel"]).all()
    assert configs.synth_queries(2, B=8).shape == (8, 7)
    assert configs.batch_size(4) == 262144 and configs.model_card(4)["out_features"] == 100
    st = configs.initial_state_from_query(configs.synth_queries(2, B=4))
    assert st.shape == (4, 7) and (st[:, [0, 1, 2, 4]] == 0).all()
