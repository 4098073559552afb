"""Checkpoint / model-card I/O: round trip in the reference's msgpack container and, where
the reference tree is mounted, decoding of the reference's own files against the committed fixture."""
import os

import numpy as np
import pytest

from conftest import DEEPER_RUN, load_ckpt_fixture, load_deeper_fixture
from irbfn_amd import checkpoint, configs
from irbfn_amd.model import WCRBFNet

REF = "/root/reference/scripts"


def test_roundtrip(tmp_path):
    P = configs.synth_params(1)
    path = checkpoint.save_checkpoint(str(tmp_path), P, step=42)
    assert os.path.basename(path) == "checkpoint_42"
    checkpoint.save_checkpoint(str(tmp_path), P, step=7)
    back, step = checkpoint.restore_checkpoint(str(tmp_path))           # directory -> latest
    assert step == 42
    for g, n in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
        np.testing.assert_array_equal(back["params"][g][n], P["params"][g][n])
    card = configs.model_card(1)
    checkpoint.save_model_card(str(tmp_path / "card.yaml"), card)
    net = WCRBFNet.from_config(str(tmp_path / "card.yaml"))
    assert net.config() == WCRBFNet.from_config(card).config()
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(FileNotFoundError):
        checkpoint.restore_checkpoint(str(empty))


def _keys(t):
    return {k: _keys(v) for k, v in t.items()} if isinstance(t, dict) else None


def test_written_tree_has_the_structure_the_reference_restores_into(tmp_path):
    """flax restores `target=state` leaf by leaf: the written opt_state must have the key structure of
    optax.chain(clip_by_global_norm, adam) -- as decoded (plain msgpack) from the reference's own newest run,
    scripts/ckpts/dnmpc_1regions_newdata_oldintloss_nomirror_highk/checkpoint_999 -- and `step` must be an array."""
    ref_opt = {"0": {}, "1": {"0": {"count": None, "mu": {"params": {"linear": {"bias": None, "kernel": None},
                                                                     "rbf_list": {"centers": None, "log_sigs": None}}},
                                    "nu": {"params": {"linear": {"bias": None, "kernel": None},
                                                      "rbf_list": {"centers": None, "log_sigs": None}}}}, "1": {}}}
    P = configs.synth_params(1)
    rng = np.random.default_rng(0)
    mu = {"params": {g: {n: rng.normal(size=a.shape).astype(np.float32) for n, a in d.items()} for g, d in P["params"].items()}}
    nu = {"params": {g: {n: np.abs(rng.normal(size=a.shape)).astype(np.float32) for n, a in d.items()} for g, d in P["params"].items()}}
    path = checkpoint.save_checkpoint(str(tmp_path), P, step=1234, opt_state=(mu, nu, 1234))
    tree = checkpoint.load_flax_msgpack(path)
    assert _keys(tree["opt_state"]) == ref_opt
    assert isinstance(tree["step"], np.ndarray) and tree["step"].shape == () and int(tree["step"]) == 1234
    assert tree["opt_state"]["1"]["0"]["count"].dtype == np.int32
    mu2, nu2, count = checkpoint.restore_opt_state(path)
    assert count == 1234
    np.testing.assert_array_equal(mu2["params"]["linear"]["kernel"], mu["params"]["linear"]["kernel"])
    np.testing.assert_array_equal(nu2["params"]["rbf_list"]["centers"], nu["params"]["rbf_list"]["centers"])
    fresh = checkpoint.load_flax_msgpack(checkpoint.save_checkpoint(str(tmp_path / "f"), P, step=0))
    assert _keys(fresh["opt_state"]) == ref_opt and not fresh["opt_state"]["1"]["0"]["mu"]["params"]["linear"]["kernel"].any()
    if os.path.isdir(REF):                    # this container: compare with the reference's own file
        ref = checkpoint.load_flax_msgpack(os.path.join(REF, "ckpts", "dnmpc_1regions_newdata_oldintloss_nomirror_highk", "checkpoint_999"))
        assert _keys(ref["opt_state"]) == ref_opt and _keys(ref["params"]) == _keys(tree["params"])
        assert isinstance(ref["step"], np.ndarray) and ref["step"].shape == ()
        old = checkpoint.restore_opt_state(os.path.join(REF, "ckpts", "dnmpc_128regions"))      # bare-adam form of older runs
        assert old is not None and old[0]["params"]["linear"]["kernel"].shape == (10, 10)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_reads_reference_checkpoint_and_card():
    run = "dnmpc_128regions"
    params, step = checkpoint.restore_checkpoint(os.path.join(REF, "ckpts", run))
    cfg, fix, *_ = load_ckpt_fixture(run)
    assert step == 799330
    np.testing.assert_array_equal(params["params"]["rbf_list"]["centers"], fix["params"]["rbf_list"]["centers"])
    np.testing.assert_array_equal(params["params"]["linear"]["kernel"], fix["params"]["linear"]["kernel"])
    card = checkpoint.load_model_card(os.path.join(REF, "configs", run + ".yaml"))
    net = WCRBFNet.from_config(card)
    assert net.num_regions == 128 and net.basis_func == "inverse_quadratic"
    with pytest.raises(ValueError):       # an MLP checkpoint is not a WCRBFNet tree
        checkpoint.restore_checkpoint(os.path.join(REF, "ckpts", "dnmpc_mlp_512"))


def test_deeper_tree_roundtrip_and_oracle(tmp_path):
    """DeeperWCRBFNet tree (model.py:254-256): save/restore keeps all four sub-trees; the oracle restatement
    reproduces the committed float64 outputs and reduces to hand-evaluated relu/Dense algebra."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import irbfn_oracle as orc
    cfg, params, x, out64 = load_deeper_fixture()
    path = checkpoint.save_checkpoint(str(tmp_path), params, 9999)
    back, step = checkpoint.restore_checkpoint(path)
    assert step == 9999 and sorted(back["params"]) == ["linear", "linear_pre1", "linear_pre2", "rbf_list"]
    for k in back["params"]:
        for n in back["params"][k]:
            np.testing.assert_array_equal(back["params"][k][n], params["params"][k][n])
    p64 = {"params": {k: {n: np.asarray(v, np.float64) for n, v in d.items()} for k, d in params["params"].items()}}
    got = orc.deeper_wcrbfnet_apply(cfg, p64, x)
    np.testing.assert_allclose(got, out64, rtol=1e-13, atol=1e-13)
    # independent algebra: stage net (WCRBFNet with linear_pre1) then two relu/Dense layers
    q = p64["params"]
    h1 = orc.wcrbfnet_apply(dict(cfg, out_features=64), {"rbf_list": q["rbf_list"], "linear": q["linear_pre1"]}, x)
    h2 = np.maximum(h1, 0) @ q["linear_pre2"]["kernel"] + q["linear_pre2"]["bias"]
    np.testing.assert_allclose(np.maximum(h2, 0) @ q["linear"]["kernel"] + q["linear"]["bias"], out64, rtol=1e-13)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_reads_reference_deeper_checkpoint():
    params, step = checkpoint.restore_checkpoint(os.path.join(REF, "ckpts", DEEPER_RUN))
    _, fix, *_ = load_deeper_fixture()
    from conftest import GOLDEN
    assert step == int(np.load(os.path.join(GOLDEN, f"ckpt_{DEEPER_RUN}.npz"))["step"]) == 2180000   # optimizer steps
    for k in ("rbf_list", "linear_pre1", "linear_pre2", "linear"):
        for n in fix["params"][k]:
            np.testing.assert_array_equal(params["params"][k][n], fix["params"][k][n])


def test_cluster_tree_roundtrip_keeps_the_gate_and_its_moments(tmp_path):
    """ClusterWCRBFNet (model.py:341-414): the gate's Dense `cluster` and its Adam moments survive save -> restore
    (they were dropped silently); an unknown parameter group is refused, not dropped."""
    rng = np.random.default_rng(3)
    R, K, D, O = 3, 8, 8, 10
    P = {"params": {"rbf_list": {"centers": rng.normal(size=(R, K, D)).astype(np.float32), "log_sigs": rng.normal(size=(R, K)).astype(np.float32)},
                    "linear": {"kernel": rng.normal(size=(K, O)).astype(np.float32), "bias": rng.normal(size=(O,)).astype(np.float32)},
                    "cluster": {"kernel": rng.normal(size=(D, R)).astype(np.float32), "bias": rng.normal(size=(R,)).astype(np.float32)}}}
    mu = {"params": {g: {n: v * 0.5 for n, v in d.items()} for g, d in P["params"].items()}}
    nu = {"params": {g: {n: v * v for n, v in d.items()} for g, d in P["params"].items()}}
    checkpoint.save_checkpoint(str(tmp_path), P, step=11, opt_state=(mu, nu, 11))
    back, step = checkpoint.restore_checkpoint(str(tmp_path))
    assert step == 11 and set(back["params"]) == {"rbf_list", "linear", "cluster"}
    for g, d in P["params"].items():
        for n, v in d.items():
            np.testing.assert_array_equal(back["params"][g][n], v)
    m2, n2, cnt = checkpoint.restore_opt_state(str(tmp_path))
    assert cnt == 11
    np.testing.assert_array_equal(m2["params"]["cluster"]["kernel"], mu["params"]["cluster"]["kernel"])
    np.testing.assert_array_equal(n2["params"]["cluster"]["bias"], nu["params"]["cluster"]["bias"])
    bad = {"params": dict(P["params"], mystery={"kernel": np.zeros((2, 2), np.float32), "bias": np.zeros(2, np.float32)})}
    with pytest.raises(ValueError):
        checkpoint.save_checkpoint(str(tmp_path), bad, step=12)
