"""float64 mode (the reference under --use_float64, scripts/train_nmpc.py:41-42): ``WCRBFNet(..., use_float64=True)``
evaluates apply / vjp in float64 on the GPU (irbfn_f64_forward / irbfn_f64_vjp).  Held to 1e-12 of the float64 oracle
(forward: the NumPy restatement; VJP: torch.autograd of the restatement), on the reference's trained checkpoints -- two of
which store float64 centers / log_sigs -- and on every basis function."""
import warnings

import numpy as np
import pytest

from conftest import CKPT_RUNS, load_ckpt_fixture
from irbfn_amd import _lib, configs
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu
LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-300))


@pytest.mark.parametrize("run", CKPT_RUNS)
def test_float64_forward_on_trained_checkpoints(gpu, run):
    cfg, P, x, out64, *_ = load_ckpt_fixture(run)
    P64 = orc.cast_params(P, np.float64)
    net = WCRBFNet.from_config(cfg, use_float64=True)
    got = net.apply(P64, x.astype(np.float64))
    assert got.dtype == np.float64 and got.shape == out64.shape
    # the committed float64 oracle outputs of the fixture; cancellation scale of the trained Dense layer (weights up to 800)
    _, h, _ = orc.wcrbfnet_apply(cfg, P64, x.astype(np.float64), return_aux=True)
    cancel = np.abs(h) @ np.abs(P64["params"]["linear"]["kernel"])
    assert (np.abs(got - out64) <= 1e-12 * np.abs(out64) + 1e-13 * cancel).all(), _rel(got, out64)
    # ragged batch sizes
    for B in (1, 63, 65):
        assert np.array_equal(net.apply(P64, x[:B].astype(np.float64)), got[:B])


@pytest.mark.parametrize("basis", sorted(_lib.BASIS_ENUM))
def test_float64_forward_and_vjp_all_bases(gpu, basis):
    import torch
    cfg = dict(configs.model_card(1), basis_func=basis)
    cfg["num_regions"] = 2                                  # two regions: the gate multiplies
    cfg["lower_bounds"] = [[1.0, 15.0], [-6.0], [-1.3]]
    cfg["upper_bounds"] = [[15.0, 30.0], [6.0], [1.3]]
    cfg["dimension_ranges"] = [[0, 0, 0], [1, 0, 0]]
    cfg["delta"] = [0.4, 15.0, 100.0]
    rng = np.random.default_rng(7)
    R, K, D, O = 2, cfg["num_kernels"], 3, cfg["out_features"]
    P = {"params": {"rbf_list": {"centers": rng.uniform([1, -6, -1.3], [30, 6, 1.3], size=(R, K, D)), "log_sigs": rng.uniform(0, 1.5, size=(R, K))},
                    "linear": {"kernel": rng.normal(size=(K, O)), "bias": rng.normal(size=(O,)) * 0.1}}}
    for B in (5, 200):
        x = rng.uniform([1, -6, -1.3], [30, 6, 1.3], size=(B, D))
        g = rng.normal(size=(B, O))
        net = WCRBFNet.from_config(cfg, use_float64=True)
        out = net.apply(P, x)
        tp = orc.torch_params(P, torch.float64, requires_grad=True)
        ref = orc.wcrbfnet_apply(cfg, tp, torch.tensor(x, dtype=torch.float64))
        assert _rel(out, ref.detach().numpy()) <= 1e-12, basis
        (ref * torch.tensor(g, dtype=torch.float64)).sum().backward()
        got = net.vjp(P, x, g)["params"]
        for grp, name in LEAVES:
            r = tp["params"][grp][name].grad.numpy()
            assert got[grp][name].dtype == np.float64
            assert _rel(got[grp][name], r) <= 1e-11, (basis, B, name, _rel(got[grp][name], r))


def test_float64_vjp_on_the_float64_checkpoint_and_the_float32_path_says_so(gpu):
    """dnmpc_128regions was trained with --use_float64 (float64 centers / log_sigs in the checkpoint): the float64 mode agrees
    with torch.autograd of the restatement to 1e-11; handing the same leaves to the float32 path warns once."""
    import torch
    cfg, P, x, *_ = load_ckpt_fixture("dnmpc_128regions")
    assert P["params"]["rbf_list"]["centers"].dtype == np.float64
    P64 = orc.cast_params(P, np.float64)
    rng = np.random.default_rng(1)
    xq = np.repeat(x.astype(np.float64), 5, axis=0) + rng.normal(size=(5 * len(x), x.shape[1])) * 0.05
    g = rng.normal(size=(len(xq), cfg["out_features"]))
    net = WCRBFNet.from_config(cfg, use_float64=True)
    got = net.vjp(P64, xq, g)["params"]
    tp = orc.torch_params(P64, torch.float64, requires_grad=True)
    ref = orc.wcrbfnet_apply(cfg, tp, torch.tensor(xq, dtype=torch.float64))
    (ref * torch.tensor(g, dtype=torch.float64)).sum().backward()
    for grp, name in LEAVES:
        r = tp["params"][grp][name].grad.numpy()
        assert _rel(got[grp][name], r) <= 1e-11, (name, _rel(got[grp][name], r))
    net32 = WCRBFNet.from_config(cfg)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        o32 = net32.apply(P, xq.astype(np.float32))
        net32.apply(P, xq.astype(np.float32))
    assert sum("float64 parameter leaves" in str(m.message) for m in w) == 1
    o64 = net.apply(P64, xq)
    _, h, _ = orc.wcrbfnet_apply(cfg, P64, xq, return_aux=True)
    cancel = np.abs(h) @ np.abs(P64["params"]["linear"]["kernel"])
    assert (np.abs(o32 - o64) <= 1e-5 * np.abs(o64) + 3e-6 * cancel).all()       # the float32 path against the float64 mode
