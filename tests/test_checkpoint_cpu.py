"""Checkpoint / model-card I/O: round trip in the reference's msgpack container and, where
the reference tree is mounted, decoding of the reference's own files against the committed fixture."""
import os

import numpy as np
import pytest

from conftest import load_ckpt_fixture
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
