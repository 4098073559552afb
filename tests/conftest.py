import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_ckpt_fixture(run):
    """(cfg, params, x, out64, h64, gamma64) of a committed checkpoint fixture."""
    z = np.load(os.path.join(GOLDEN, f"ckpt_{run}.npz"))
    cfg = json.load(open(os.path.join(GOLDEN, f"ckpt_{run}.json")))
    params = {"params": {"rbf_list": {"centers": z["centers"], "log_sigs": z["log_sigs"]},
                         "linear": {"kernel": z["kernel"], "bias": z["bias"]}}}
    return cfg, params, z["x"], z["out64"], z["h64"], z["gamma64"]


DEEPER_RUN = "dnmpc_1regions_frenet_l1_bigdata_5stepint_deeper"


def load_deeper_fixture(run=DEEPER_RUN):
    """(cfg, params, x, out64) of the committed DeeperWCRBFNet checkpoint fixture."""
    z = np.load(os.path.join(GOLDEN, f"ckpt_{run}.npz"))
    cfg = json.load(open(os.path.join(GOLDEN, f"ckpt_{run}.json")))
    params = {"params": {k: {n: z[f"{k}__{n}"] for n in names} for k, names in (
        ("rbf_list", ("centers", "log_sigs")), ("linear_pre1", ("kernel", "bias")),
        ("linear_pre2", ("kernel", "bias")), ("linear", ("kernel", "bias")))}}
    return cfg, params, z["x"], z["out64"]


CKPT_RUNS = ["dnmpc_1regions_newdata_oldintloss_nomirror_highk", "dnmpc_128regions",
             "dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq", "dnmpc_12regions_frenet_l1_bigdata"]


@pytest.fixture(scope="session")
def kat():
    return json.load(open(os.path.join(GOLDEN, "kat.json")))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch
