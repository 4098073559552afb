"""Plausibility pin of the UNPINNED RBF stage.  The reference holds no recorded output of
any RBFLayer / WCRBFNet call, so the restatement in oracle/irbfn_oracle.py is checked against the only reference-
held evidence there is: its TRAINED checkpoints.  A net trained to imitate NMPC solutions must, on queries inside
its training range, return controls of the size the vehicle limits allow (dynamics.py:46-47 clips to
|a| <= 9.51, |sv| <= 3.2; dyn_params of scripts/test_dynamics.ipynb cell 1) -- with Dense weights up to 480 that
only happens if widths, distances and the gate are read the way the nets were trained.

What this excludes (each misreading moves the controls by 4x to 100x): sigma^2 where sigma belongs
(d = r / sigma, flax_rbf.py:280), the squared distance handed to a basis that squares again (phi(d) = exp(-d^2),
:35-37), and dropping the region gate on a multi-region net (model.py:187-193).
What it does NOT exclude (outputs stay in range under them; documented in DESIGN.md section 2): 1/sigma in place
of sigma, and r^2/sigma in place of sqrt(r^2)/sigma.  It does not lift "parity unpinned"."""
import numpy as np
import pytest

from conftest import load_ckpt_fixture
from oracle import irbfn_oracle as orc

A_MAX, SV_MAX = 9.51, 3.2           # dyn_params[10], dyn_params[9]
CARTESIAN = ["dnmpc_1regions_newdata_oldintloss_nomirror_highk", "dnmpc_128regions",
             "dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq"]
ALL = CARTESIAN + ["dnmpc_12regions_frenet_l1_bigdata"]


def _variants(run):
    cfg, P, x, out64, h64, g64 = load_ckpt_fixture(run)
    p = orc.cast_params(P, np.float64)["params"]
    c, ls, W, b = p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"], p["linear"]["kernel"], p["linear"]["bias"]
    r2 = ((x[:, None, None, :].astype(np.float64) - c[None]) ** 2).sum(-1)
    basis = orc.BASIS[cfg["basis_func"]]

    def net(d, gated=True):
        phi = basis(d, orc._NP)
        return ((g64[:, :, None] * phi).sum(1) if gated else phi.sum(1)) @ W + b
    T = cfg["out_features"] // 2
    outs = {"restatement": net(np.sqrt(r2) / np.exp(ls)[None]),
            "sigma_squared": net(np.sqrt(r2) / np.exp(2 * ls)[None]),
            "d_squared_into_basis": net(r2 / np.exp(2 * ls)[None]),
            "no_gate": net(np.sqrt(r2) / np.exp(ls)[None], gated=False)}
    np.testing.assert_allclose(outs["restatement"], out64, rtol=1e-9, atol=1e-9)    # the fixture IS the restatement
    return cfg, {k: (np.abs(v[:, :T]), np.abs(v[:, T:])) for k, v in outs.items()}


@pytest.mark.parametrize("run", CARTESIAN)
def test_trained_cartesian_nets_return_controls_inside_the_vehicle_limits(run):
    cfg, v = _variants(run)
    a, sv = v["restatement"]
    # observed: max |a| 8.0 ... 11.6, 90 % of the accelerations below 7.3; max |sv| 7.2 ... 8.9, 90 % below 4.8
    assert a.max() <= 1.25 * A_MAX and np.quantile(a, 0.9) <= A_MAX, (a.max(), np.quantile(a, 0.9))
    assert sv.max() <= 3.0 * SV_MAX and np.quantile(sv, 0.9) <= 1.5 * SV_MAX, (sv.max(), np.quantile(sv, 0.9))
    assert np.median(a) >= 0.1 * A_MAX                       # and they are not collapsed to the bias either


@pytest.mark.parametrize("run", ALL)
def test_misread_conventions_leave_the_physical_range(run):
    cfg, v = _variants(run)
    a0 = np.median(v["restatement"][0])
    for name in ("sigma_squared", "d_squared_into_basis"):
        a, sv = v[name]
        assert np.median(a) >= 4.0 * a0, (name, np.median(a), a0)
        assert np.quantile(a, 0.9) > 2.5 * A_MAX, (name, np.quantile(a, 0.9))
    if cfg["num_regions"] > 1:                                # the gate is what partitions the regions
        a, sv = v["no_gate"]
        assert np.median(a) >= 10.0 * a0 and np.quantile(a, 0.9) > 10 * A_MAX
