"""GPU parity of K1g (`rbf_fwd_f16gram` / `rbf_tick_f16gram`, irbfn_amd/csrc/rbf_forward_gram.hip): the narrow forward with the
squared distances as a Gram expansion on the f16 matrix cores (exact fixed-point heads + float tails) in front of K1h's
Phi x W.  Against the float64 oracle scaled by sum_k |phi_k W_k| (the natural error scale of the reduction), against the
all-float32 kernel K1 and K1h, at 1e-5 relative to |ref| on ill-conditioned columns built without cancellation; queries
outside the representable box (and non-finite ones) take K1h's distances wave by wave; parameters outside the exactness budget
are refused."""
import numpy as np
import pytest

from conftest import load_ckpt_fixture
from irbfn_amd import _lib, configs
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu

DEFAULTS = {"fwd_kernel": 0, "fwd_f16_s": 0, "fwd_f16_qg": 0}


def _run(net, params, x, kernel=None, **opts):
    net.set_options(fwd_kernel=_lib.FWD_K1G if kernel is None else kernel, **opts)
    try:
        got = net.apply(params, x)
        name = net.last_launch()["kernel"]
    finally:
        net.set_options(**{k: DEFAULTS[k] for k in ["fwd_kernel", *opts]})
    return got, name


def _terms_scale(cfg, p64, x):
    pa = {"params": {"rbf_list": p64["params"]["rbf_list"],
                     "linear": {"kernel": np.abs(p64["params"]["linear"]["kernel"]), "bias": np.abs(p64["params"]["linear"]["bias"])}}}
    return orc.wcrbfnet_apply(cfg, pa, x)


def _card(D, K, O, basis, lo, hi, delta=20.0):
    return {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": basis, "num_regions": 1,
            "lower_bounds": [[float(v)] for v in lo], "upper_bounds": [[float(v)] for v in hi],
            "dimension_ranges": [[0] * D], "activation_idx": list(range(D)), "delta": [delta] * D}


def test_gram_is_the_default_at_config_2_and_as_accurate_as_the_float32_kernel(gpu):
    cfg, params = configs.model_card(2), configs.synth_params(2)
    net = WCRBFNet.from_config(cfg)
    B = 4 * 4096 + 37                                # ragged tail; the automatic choice takes K1g from 12288 queries
    x = configs.synth_queries(2, B=B)
    auto = net.apply(params, x)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram<D=7,BC=0"), net.last_launch()
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16gram<") and np.array_equal(got, auto)
    p64 = orc.cast_params(params, np.float64)
    x64 = x.astype(np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x64)
    scale = _terms_scale(cfg, p64, x64)
    err = np.abs(got - ref) / scale
    k1, nm = _run(net, params, x, kernel=_lib.FWD_K1)
    assert nm.startswith("rbf_fwd_qlane")
    k1h, nm = _run(net, params, x, kernel=_lib.FWD_K1H)
    assert nm.startswith("rbf_fwd_f16mfma<")
    err_k1, err_h = np.abs(k1 - ref) / scale, np.abs(k1h - ref) / scale
    print(f"K1g max/mean err {err.max():.2e}/{err.mean():.2e}   K1h {err_h.max():.2e}/{err_h.mean():.2e}   K1 {err_k1.max():.2e}/{err_k1.mean():.2e}")
    # float32-equivalent: no worse than the float32 FMA-chain kernel on the same inputs
    # (the mean carries the truncation of the tail products on top: measured 1.9e-8 against 1.15e-8 for K1, 1.24e-8 for K1h)
    assert err.max() <= 3e-6 and err.max() <= 2.0 * err_k1.max() and err.mean() <= 2.5 * err_k1.mean(), (err.max(), err_k1.max())
    assert (np.abs(got - k1) / scale).max() <= 1e-5
    # every (S, QG) geometry gives the same answer up to the slice summation order; the answer does not depend on the run
    for S, QG in ((1, 8), (2, 4), (2, 8), (1, 4), (4, 1), (1, 1)):
        g2, nm = _run(net, params, x, fwd_f16_s=S, fwd_f16_qg=QG)
        assert f"S={S},QG={QG}" in nm
        assert (np.abs(g2 - ref) / scale).max() <= 3e-6, (S, QG)
        g3, _ = _run(net, params, x, fwd_f16_s=S, fwd_f16_qg=QG)
        assert np.array_equal(g2, g3)


@pytest.mark.parametrize("B", [65, 100, 2500])
def test_gram_on_the_reference_s_trained_single_region_planner(gpu, B):
    """Trained single-region net of the reference (gaussian, O = 10; K = 1000 is not a multiple of the 32-centre chunk; bounds
    make gamma != 1 near the edges)."""
    cfg, params, x0, out64, *_ = load_ckpt_fixture("dnmpc_1regions_newdata_oldintloss_nomirror_highk")
    net = WCRBFNet.from_config(cfg)
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)])
    hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = np.random.default_rng(B).uniform(lo - 0.02, hi + 0.02, size=(B, cfg["in_features"])).astype(np.float32)
    x[:64] = x0.astype(np.float32)
    p32 = orc.cast_params(params, np.float32)
    got, name = _run(net, p32, x)
    assert name.startswith("rbf_fwd_f16gram<"), name
    p64 = orc.cast_params(p32, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
    assert (np.abs(got - ref) / scale).max() <= 2e-6
    gam = orc.region_activation(x.astype(np.float64), cfg["num_regions"], len(cfg["activation_idx"]), cfg["lower_bounds"],
                                cfg["upper_bounds"], cfg["delta"], cfg["dimension_ranges"])
    assert gam.min() < 0.9 and gam.max() > 0.99      # the gate is exercised


@pytest.mark.parametrize("D,K,O,basis", [(3, 256, 5, "gaussian"), (4, 96, 16, "inverse_multiquadric"), (7, 200, 2, "inverse_quadratic"),
                                         (7, 33, 1, "gaussian_wide"), (5, 64, 7, "inverse_multiquadric"), (2, 50, 3, "gaussian_wider"),
                                         (6, 1000, 10, "gaussian"), (7, 20, 4, "gaussian"), (4, 32, 16, "inverse_quadratic"),
                                         (8, 300, 2, "gaussian"), (8, 100, 10, "inverse_multiquadric")])       # d = 8: the Frenet nets
def test_gram_shapes_and_bases(gpu, D, K, O, basis):
    rng = np.random.default_rng(D * 100 + K)
    lo, hi = -np.ones(D) * 2, np.ones(D) * 3
    cfg = _card(D, K, O, basis, lo, hi)
    params = {"params": {"rbf_list": {"centers": rng.uniform(lo - 1, hi + 1, size=(1, K, D)).astype(np.float32),
                                      "log_sigs": rng.uniform(-0.5, 1.5, size=(1, K)).astype(np.float32)},
                         "linear": {"kernel": (rng.normal(size=(K, O)) * rng.choice([1e-3, 1.0, 300.0], size=(1, O))).astype(np.float32),
                                    "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    for B in (1000, 70):
        x = rng.uniform(lo, hi, size=(B, D)).astype(np.float32)
        got, name = _run(net, params, x)
        assert name.startswith("rbf_fwd_f16gram<"), name
        p64 = orc.cast_params(params, np.float64)
        ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
        scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
        assert (np.abs(got - ref) / scale).max() <= 2e-6, B


@pytest.mark.parametrize("basis", ["gaussian", "inverse_quadratic", "inverse_multiquadric"])
@pytest.mark.parametrize("case", ["outlier_far_centre", "six_decades", "queries_on_small_weight_centres", "columns_1e-6_to_1e6"])
def test_gram_ill_conditioned_columns(gpu, case, basis):
    """The adversarial columns of tests/test_gpu_f16.py::test_forward_f16_ill_conditioned_columns on K1g: positive weights (no
    cancellation: |ref| is the error scale), 1e-5 relative to |ref| and float32-grade (as good as the all-float32 K1).  The
    queries sitting ON centres make the expansion cancel completely -- the case the exact head sum is for.  (With the gaussian
    the far outlier centre at 60 widens the box of the expansion past its budget: that net stays on K1h.)"""
    from test_gpu_f16 import _cond_net
    K, O = 512, 10
    rng, cfg, centers, log_sigs = _cond_net(K, O, basis, seed=len(case))
    B = 2048 + 5
    x = rng.uniform(0.0, 4.0, size=(B, 7)).astype(np.float32)
    W = np.abs(rng.normal(size=(K, O))) + 0.05
    if case == "outlier_far_centre":
        centers[0, 3] = 60.0
        log_sigs[0, 3] = -0.5
        W[3, :] = 1.0e4
    elif case == "six_decades":
        W = 10.0 ** rng.uniform(-3, 3, size=(K, O))
    elif case == "queries_on_small_weight_centres":
        W = 10.0 ** rng.uniform(-3, 3, size=(K, O))
        small = np.argsort(W.max(axis=1))[:64]
        x[:1024] = centers[0, small[rng.integers(0, 64, size=1024)]] + rng.normal(0, 0.02, size=(1024, 7)).astype(np.float32)
        x[:256] = centers[0, small[rng.integers(0, 64, size=256)]]          # exactly ON a centre
        log_sigs[0, small] = -1.5
    else:
        W = W * 10.0 ** np.linspace(-6, 6, O)[None, :]
    params = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs},
                         "linear": {"kernel": W.astype(np.float32), "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    if case == "outlier_far_centre" and basis == "gaussian":     # 2 alpha c' leaves the budget: refused, the net stays on K1h
        with pytest.raises(ValueError, match="UNSUPPORTED"):
            _run(net, params, x)
        got, name = _run(net, params, x, kernel=_lib.FWD_AUTO)
        assert name.startswith("rbf_fwd_f16mfma<"), name
    else:
        got, name = _run(net, params, x)
        assert name.startswith("rbf_fwd_f16gram<"), name
    k1, _ = _run(net, params, x, kernel=_lib.FWD_K1)
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x.astype(np.float64))
    assert (ref > 0).all()
    rel, rel_k1 = np.abs(got - ref) / ref, np.abs(k1 - ref) / ref
    print(f"{case}/{basis}: {name.split('<')[0]} max rel {rel.max():.2e} (K1 {rel_k1.max():.2e}), median {np.median(rel):.2e}")
    assert rel.max() <= 1e-5, (case, basis, rel.max(), rel_k1.max())        # north-star: 1e-5 relative to |ref|
    assert rel.max() <= max(3e-6, 2.5 * rel_k1.max())                        # and float32-grade: as good as K1


def test_gram_narrow_widths_far_from_the_origin(gpu):
    """What the exact head sum is for: centres and queries far from the expansion's origin (|c'| up to 6 in units where the
    widths are 0.45 .. 1), queries within a fraction of a width of a centre.  The terms of the expansion reach 2^10 here while u
    stays below 1: a float32-accumulated expansion would be off by 2^-24 * 2^10 = 6e-5 in u; the result must stay at the
    float32 kernel's level."""
    rng = np.random.default_rng(5)
    D, K, O, B = 7, 256, 4, 4096
    lo, hi = -6.0 * np.ones(D), 6.0 * np.ones(D)
    cfg = _card(D, K, O, "gaussian", lo, hi, delta=100.0)
    centers = rng.uniform(lo, hi, size=(1, K, D)).astype(np.float32)
    centers[0, :16] = np.sign(rng.normal(size=(16, D))) * rng.uniform(5.5, 6.0, size=(16, D))        # corners
    log_sigs = rng.uniform(-0.8, 0.0, size=(1, K)).astype(np.float32)
    W = (np.abs(rng.normal(size=(K, O))) + 0.1).astype(np.float32)
    x = (centers[0, rng.integers(0, K, size=B)] + rng.normal(0, 0.15, size=(B, D))).astype(np.float32)
    x = np.clip(x, lo + 0.05, hi - 0.05).astype(np.float32)
    params = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs}, "linear": {"kernel": W, "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16gram<"), name
    k1, _ = _run(net, params, x, kernel=_lib.FWD_K1)
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x.astype(np.float64))
    keep = ref > 1e-3                                                                                  # outputs that see a centre
    rel, rel_k1 = (np.abs(got - ref) / ref)[keep], (np.abs(k1 - ref) / ref)[keep]
    print(f"far from the origin: K1g max rel {rel.max():.2e} mean {rel.mean():.2e}   K1 {rel_k1.max():.2e} mean {rel_k1.mean():.2e}")
    assert keep.mean() > 0.5 and rel.max() <= 1e-5 and rel.max() <= 10.0 * rel_k1.max() + 3e-6


def test_gram_queries_outside_the_box_take_the_float32_distances(gpu):
    """A wave with a query outside the representable box (here: 1e3 and 1e6 away, +-Inf, NaN) computes its 32 queries' distances
    on the VALU from K1h's records (K1h's arithmetic; the summation order over the centres is K1g's): NaN propagates like jnp,
    Inf gives bias (gamma = 0, phi = 0), every other wave is untouched."""
    cfg, params = configs.model_card(2), configs.synth_params(2)
    net = WCRBFNet.from_config(cfg)
    B = 32 * 40 + 5
    x = configs.synth_queries(2, B=B)
    clean, _ = _run(net, params, x)
    xb = x.copy()
    xb[3, 2] = 1.0e3
    xb[40, 0] = -1.0e6
    xb[70, 5] = np.inf
    xb[100, 1] = -np.inf
    xb[130, 6] = np.nan
    xb[B - 1, 3] = 50.0                                   # in the ragged tail group
    got, name = _run(net, params, xb)
    assert name.startswith("rbf_fwd_f16gram<")
    k1h, _ = _run(net, params, xb, kernel=_lib.FWD_K1H)
    bad_groups = sorted({r // 32 for r in (3, 40, 70, 100, 130, B - 1)})
    rows_bad = np.concatenate([np.arange(g * 32, min((g + 1) * 32, B)) for g in bad_groups])
    rows_ok = np.setdiff1d(np.arange(B), rows_bad)
    assert np.array_equal(got[rows_ok], clean[rows_ok])
    p64 = orc.cast_params(params, np.float64)
    fin = rows_bad[np.isfinite(xb[rows_bad]).all(axis=1)]
    ref = orc.wcrbfnet_apply(cfg, p64, xb[fin].astype(np.float64))
    scale = _terms_scale(cfg, p64, xb[fin].astype(np.float64)) + 1e-30
    assert (np.abs(got[fin] - ref) / scale).max() <= 3e-6 and (np.abs(got[fin] - k1h[fin]) / scale).max() <= 3e-6
    assert np.isnan(got[130]).all() and np.isfinite(got[[3, 40, 70, 100, B - 1]]).all()
    bias = np.asarray(params["params"]["linear"]["bias"], np.float32)
    assert np.allclose(got[70], bias, atol=1e-6) and np.allclose(got[100], bias, atol=1e-6)


def test_gram_refuses_parameters_outside_its_budget(gpu):
    """Widths of 1e-3 next to centres spread over +-5: alpha |c'|^2 ~ 1e7 leaves the f16 operand range of the expansion.  The
    pack says so (read back by irbfn_net_set_params), the automatic choice stays on K1h, forcing K1g is refused."""
    rng = np.random.default_rng(9)
    D, K, O = 7, 128, 10
    lo, hi = -5.0 * np.ones(D), 5.0 * np.ones(D)
    cfg = _card(D, K, O, "gaussian", lo, hi)
    params = {"params": {"rbf_list": {"centers": rng.uniform(lo, hi, size=(1, K, D)).astype(np.float32),
                                      "log_sigs": np.full((1, K), -7.0, np.float32)},
                         "linear": {"kernel": rng.normal(size=(K, O)).astype(np.float32), "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    x = rng.uniform(lo, hi, size=(500, D)).astype(np.float32)
    x = np.tile(x, (30, 1))                               # 15000 queries: the automatic choice would take K1g
    auto = net.apply(params, x)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16mfma<"), net.last_launch()
    with pytest.raises(ValueError, match="UNSUPPORTED"):
        _run(net, params, x)
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x.astype(np.float64))
    assert np.abs(auto - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30) + 1e-12
    # NaN parameters: refused as well (the statistics of the pack are not finite)
    params["params"]["rbf_list"]["centers"][0, 3, 2] = np.nan
    net2 = WCRBFNet.from_config(cfg)
    net2.apply(params, x)
    assert not net2.last_launch()["kernel"].startswith("rbf_fwd_f16gram")


def test_gram_multi_region_nets_are_not_eligible(gpu):
    cfg, params, x, *_ = load_ckpt_fixture("dnmpc_128regions")
    net = WCRBFNet.from_config(cfg)
    with pytest.raises(ValueError, match="UNSUPPORTED"):
        _run(net, orc.cast_params(params, np.float32), x.astype(np.float32))


# ---- wide outputs (16 < O <= 128): rbf_fwd_f16gram_wide, rbf_forward_gram_wide.h ------------------------------------------
@pytest.mark.parametrize("O,SW,basis", [(100, 2, "gaussian"), (100, 1, "gaussian"), (20, 4, "inverse_quadratic"), (64, 1, "inverse_multiquadric"),
                                        (128, 2, "gaussian"), (33, 0, "gaussian")])
def test_gram_wide_outputs(gpu, O, SW, basis):
    """50-step control sequences (O = 100) and the other column-tile counts: K1g's wide form against float64, the f32 matrix-core
    kernel K1m and K1h's wide kernel."""
    rng = np.random.default_rng(O + SW)
    D, K = 7, 300
    cfg = dict(configs.model_card(4), num_kernels=K, out_features=O, basis_func=basis)
    params = {"params": {"rbf_list": {"centers": rng.uniform(-1, 8, size=(1, K, D)).astype(np.float32),
                                      "log_sigs": rng.uniform(0.0, 2.0, size=(1, K)).astype(np.float32)},
                         "linear": {"kernel": rng.normal(0, 0.3, size=(K, O)).astype(np.float32),
                                    "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    for B in (777, 2500):
        x = configs.synth_queries(4, B=B)
        got, name = _run(net, params, x, fwd_f16_s=SW)
        assert name.startswith("rbf_fwd_f16gram_wide<") and (SW == 0 or f"SW={min(SW, 2 if O > 48 else 4)}" in name), name
        p64 = orc.cast_params(params, np.float64)
        ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
        scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
        assert got.shape == (B, O)
        other, nm = _run(net, params, x, kernel=_lib.FWD_K1M)
        assert nm.startswith("rbf_fwd_mfma")
        err, err_other = (np.abs(got - ref) / scale).max(), (np.abs(other - ref) / scale).max()
        assert err <= max(2e-6, 2.0 * err_other), (err, err_other)
        assert (np.abs(got - other) / scale).max() <= 1e-5
        auto = net.apply(params, x)                           # the automatic choice takes it from 2048 queries
        assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram_wide<" if B >= 2048 else "rbf_fwd_f16mfma_wide"), net.last_launch()
        assert (np.abs(auto - ref) / scale).max() <= max(3e-6, 2.0 * err_other)


@pytest.mark.parametrize("O,case", [(100, "six_decades"), (40, "six_decades"), (100, "on_centres")])
def test_gram_wide_ill_conditioned_columns(gpu, O, case):
    from test_gpu_f16 import _cond_net
    K = 300
    rng, cfg, centers, log_sigs = _cond_net(K, O, "gaussian", seed=O)
    B = 1024 + 3
    x = rng.uniform(0.0, 4.0, size=(B, 7)).astype(np.float32)
    W = 10.0 ** rng.uniform(-3, 3, size=(K, O))
    if case == "on_centres":
        x[:512] = centers[0, rng.integers(0, K, size=512)]
    params = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs},
                         "linear": {"kernel": W.astype(np.float32), "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16gram_wide<"), name
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x.astype(np.float64))
    rel = np.abs(got - ref) / ref
    print(f"wide {case}/O={O}: max rel {rel.max():.2e}")
    assert rel.max() <= 1e-5, (case, O, rel.max())


def test_gram_wide_d8_like_the_deeper_frenet_stage(gpu):
    """d = 8, O = 64: the RBF stage + linear_pre1 of the reference's DeeperWCRBFNet Frenet planner (model.py:201-289) as one wide net."""
    rng = np.random.default_rng(64)
    D, K, O = 8, 300, 64
    cfg = _card(D, K, O, "gaussian", [-1.0] * D, [2.0] * D)
    params = {"params": {"rbf_list": {"centers": rng.uniform(-1.5, 2.5, size=(1, K, D)).astype(np.float32),
                                      "log_sigs": rng.uniform(0.0, 1.0, size=(1, K)).astype(np.float32)},
                         "linear": {"kernel": rng.normal(0, 0.3, size=(K, O)).astype(np.float32), "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    x = rng.uniform(-1.0, 2.0, size=(2500, D)).astype(np.float32)
    got = net.apply(params, x)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram_wide<D=8"), net.last_launch()
    p64 = orc.cast_params(params, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
    assert (np.abs(got - ref) / scale).max() <= 3e-6


def test_gram_wide_queries_outside_the_box(gpu):
    cfg, params = configs.model_card(4), configs.synth_params(4)
    net = WCRBFNet.from_config(cfg)
    B = 32 * 70 + 9
    x = configs.synth_queries(4, B=B)
    clean, _ = _run(net, params, x)
    xb = x.copy()
    xb[5, 1] = 4.0e2
    xb[77, 0] = np.inf
    xb[200, 6] = np.nan
    got, name = _run(net, params, xb)
    assert name.startswith("rbf_fwd_f16gram_wide<")
    k1h, _ = _run(net, params, xb, kernel=_lib.FWD_K1H)
    rows_bad = np.concatenate([np.arange(g * 32, (g + 1) * 32) for g in sorted({5 // 32, 77 // 32, 200 // 32})])
    rows_ok = np.setdiff1d(np.arange(B), rows_bad)
    assert np.array_equal(got[rows_ok], clean[rows_ok])
    fin = rows_bad[np.isfinite(xb[rows_bad]).all(axis=1)]
    p64 = orc.cast_params(params, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, xb[fin].astype(np.float64))
    scale = _terms_scale(cfg, p64, xb[fin].astype(np.float64)) + 1e-30
    assert (np.abs(got[fin] - ref) / scale).max() <= 3e-6 and (np.abs(got[fin] - k1h[fin]) / scale).max() <= 3e-6
    assert np.isnan(got[200]).all() and np.isfinite(got[[5, 77]]).all()
    assert np.allclose(got[77], np.asarray(params["params"]["linear"]["bias"], np.float32), atol=1e-6)


# ---- K2g: the parameter VJP with u and the centre gradients on the matrix cores as well (rbf_vjp_gram.hip) -----------------
LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


def _vjp_case(case, rng):
    if case == "cfg3":
        cfg = dict(configs.model_card(3), num_kernels=500)                 # not a multiple of 32 / 128: idle waves in the last block
        P = configs.synth_params(3)
        P = {"params": {"rbf_list": {k: v[:, :500] for k, v in P["params"]["rbf_list"].items()},
                        "linear": {"kernel": P["params"]["linear"]["kernel"][:500], "bias": P["params"]["linear"]["bias"]}}}
        B = 4096 + 19
        x, g = configs.synth_queries(3, B=B), configs.synth_cotangent(3, B=B) * 1e-3
    elif case == "iq_ckpt":
        cfg, P, *_ = load_ckpt_fixture("dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq")
        P = orc.cast_params(P, np.float32)
        ns = len(cfg["activation_idx"])
        lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
        B = 3000
        x = rng.uniform(lo - 0.02, hi + 0.02, size=(B, 7)).astype(np.float32)
        g = rng.normal(size=(B, 2)).astype(np.float32) * 40.0
    elif case == "gauss_ckpt":
        cfg, P, *_ = load_ckpt_fixture("dnmpc_1regions_newdata_oldintloss_nomirror_highk")
        P = orc.cast_params(P, np.float32)
        ns = len(cfg["activation_idx"])
        lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
        B = 2500
        x = rng.uniform(lo - 0.02, hi + 0.02, size=(B, 7)).astype(np.float32)
        g = rng.normal(size=(B, 10)).astype(np.float32)
    elif case == "iq_o16":
        D, K, O = 7, 300, 16                                                 # more than 10 outputs: hbar as two MFMAs (A1 + A2)
        cfg = _card(D, K, O, "inverse_quadratic", [-1.0] * D, [2.0] * D)
        P = {"params": {"rbf_list": {"centers": rng.uniform(-1.5, 2.5, size=(1, K, D)).astype(np.float32),
                                     "log_sigs": rng.uniform(-0.3, 0.8, size=(1, K)).astype(np.float32)},
                        "linear": {"kernel": (rng.normal(size=(K, O)) * 10.0 ** rng.uniform(-2, 1, size=(1, O))).astype(np.float32),
                                   "bias": rng.normal(size=(O,)).astype(np.float32)}}}
        B = 2500
        x = rng.uniform(-1.1, 2.1, size=(B, D)).astype(np.float32)
        g = rng.normal(size=(B, O)).astype(np.float32)
    elif case == "gauss_d8":
        D, K, O = 8, 200, 2                                                  # the Frenet planners' width
        cfg = _card(D, K, O, "gaussian", [-1.0] * D, [2.0] * D)
        P = {"params": {"rbf_list": {"centers": rng.uniform(-1.5, 2.5, size=(1, K, D)).astype(np.float32),
                                     "log_sigs": rng.uniform(0.0, 1.0, size=(1, K)).astype(np.float32)},
                        "linear": {"kernel": rng.normal(size=(K, O)).astype(np.float32), "bias": rng.normal(size=(O,)).astype(np.float32)}}}
        B = 3000
        x = rng.uniform(-1.1, 2.1, size=(B, D)).astype(np.float32)
        g = rng.normal(size=(B, O)).astype(np.float32)
    else:
        D, K, O = 3, 96, 5
        cfg = _card(D, K, O, "inverse_multiquadric", [-2.0] * D, [3.0] * D)
        P = {"params": {"rbf_list": {"centers": rng.uniform(-3, 4, size=(1, K, D)).astype(np.float32),
                                     "log_sigs": rng.uniform(-0.5, 1.0, size=(1, K)).astype(np.float32)},
                        "linear": {"kernel": (rng.normal(size=(K, O)) * np.array([1e-2, 1, 1, 50, 1])).astype(np.float32),
                                   "bias": rng.normal(size=(O,)).astype(np.float32)}}}
        B = 2048 + 1
        x = rng.uniform(-2.2, 3.2, size=(B, D)).astype(np.float32)
        g = rng.normal(size=(B, O)).astype(np.float32)
        g[7] = 0.0
    return cfg, P, x, g


@pytest.mark.parametrize("case", ["cfg3", "iq_ckpt", "gauss_ckpt", "imq_d3", "gauss_d8", "iq_o16"])
def test_vjp_gram_matches_valu_kernel_and_oracle(gpu, case):
    import torch
    cfg, P, x, g = _vjp_case(case, np.random.default_rng(11))
    net = WCRBFNet.from_config(cfg)
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    net.set_options(vjp_kernel=_lib.VJP_K2G)
    a = net.vjp(P, xt, gt)["params"]
    a2 = net.vjp(P, xt, gt)["params"]
    net.set_options(vjp_kernel=_lib.VJP_K2)
    b = net.vjp(P, xt, gt)["params"]
    net.set_options(vjp_kernel=_lib.VJP_AUTO)
    ref = orc.wcrbfnet_vjp(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), g.astype(np.float64))["params"]
    for grp, name in LEAVES:
        ga, gb, gr = a[grp][name].cpu().numpy(), b[grp][name].cpu().numpy(), np.asarray(ref[grp][name])
        assert torch.equal(a[grp][name], a2[grp][name])                    # deterministic
        scale = np.abs(gr).max() + 1e-30
        ea, eb = np.abs(ga - gr).max() / scale, np.abs(gb - gr).max() / scale
        print(f"{case} {name}: K2g {ea:.2e}  K2 {eb:.2e}")
        assert ea <= max(2e-5, 2.0 * eb), (case, grp, name, ea, eb)


@pytest.mark.parametrize("O", [5, 16])
@pytest.mark.parametrize("basis", ["gaussian", "inverse_quadratic", "inverse_multiquadric"])
@pytest.mark.parametrize("D", [3, 4, 7, 8])
def test_vjp_gram_every_instance_is_deterministic_and_agrees_with_the_float32_kernel(gpu, D, basis, O):
    """Every compiled instance of K2g (4 widths x 3 basis classes x hbar as one / two MFMAs): bit-identical run to run and equal
    to the all-float32 K2 to 2e-5 of each leaf's largest entry.  (What found the MFMA hazard of DESIGN section 4, K2g: a 16x16x32
    MFMA accumulating onto a 16x16x16 result is not interlocked on gfx950; two faster builds of this kernel returned garbage in 24 and
    in 4 of these instances while the narrower case list above passed all but one of them.)"""
    import torch
    rng = np.random.default_rng(100 * D + O)
    K, B = 200, 2100
    cfg = _card(D, K, O, basis, [-1.0] * D, [2.0] * D)
    P = {"params": {"rbf_list": {"centers": rng.uniform(-1.4, 2.4, size=(1, K, D)).astype(np.float32),
                                 "log_sigs": rng.uniform(-0.2, 0.9, size=(1, K)).astype(np.float32)},
                    "linear": {"kernel": rng.normal(size=(K, O)).astype(np.float32), "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    x = torch.from_numpy(rng.uniform(-1.05, 2.05, size=(B, D)).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.normal(size=(B, O)).astype(np.float32)).cuda()
    net = WCRBFNet.from_config(cfg)
    net.set_options(vjp_kernel=_lib.VJP_K2G)
    runs = [net.vjp(P, x, g)["params"] for _ in range(3)]
    assert net.last_launch()["kernel"].startswith("rbf_vjp_f16gram<"), net.last_launch()
    net.set_options(vjp_kernel=_lib.VJP_K2)
    ref = net.vjp(P, x, g)["params"]
    net.set_options(vjp_kernel=_lib.VJP_AUTO)
    for grp, name in LEAVES:
        a0 = runs[0][grp][name]
        assert torch.isfinite(a0).all(), (D, basis, O, name)
        assert torch.equal(a0, runs[1][grp][name]) and torch.equal(a0, runs[2][grp][name]), (D, basis, O, name)
        r = ref[grp][name]
        err = float((a0 - r).abs().max() / (r.abs().max() + 1e-30))
        assert err <= 2e-5, (D, basis, O, name, err)


def test_vjp_gram_hands_over_when_a_query_leaves_the_box(gpu):
    """A query outside the representable box of the expansion raises the pre-pass's flag: K2g returns at once and K2h, launched
    behind it, computes the slabs -- the gradients are right either way (here against the float64 restatement)."""
    import torch
    cfg, P, x, g = _vjp_case("cfg3", np.random.default_rng(3))
    x = x.copy()
    x[100, 2] = 300.0                                      # far outside: gamma = 0, contributes nothing, cannot be expanded
    x[2000, 0] = -1.0e4
    net = WCRBFNet.from_config(cfg)
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    net.set_options(vjp_kernel=_lib.VJP_K2G)
    a = net.vjp(P, xt, gt)["params"]
    net.set_options(vjp_kernel=_lib.VJP_AUTO)
    ref = orc.wcrbfnet_vjp(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), g.astype(np.float64))["params"]
    for grp, name in LEAVES:
        gr = np.asarray(ref[grp][name])
        assert np.abs(a[grp][name].cpu().numpy() - gr).max() <= 2e-5 * (np.abs(gr).max() + 1e-30), (grp, name)


def test_vjp_gram_is_refused_where_the_expansion_is(gpu):
    import torch
    cfg, P, x, *_ = load_ckpt_fixture("dnmpc_128regions")
    net = WCRBFNet.from_config(cfg)
    net.set_options(vjp_kernel=_lib.VJP_K2G)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    with pytest.raises(ValueError, match="UNSUPPORTED"):
        net.vjp(orc.cast_params(P, np.float32), xt, torch.ones(x.shape[0], cfg["out_features"], device="cuda"))
    net.set_options(vjp_kernel=_lib.VJP_AUTO)


def test_gram_sticky_verdict_costs_speed_never_correctness(gpu):
    """IRBFN_OPT_GRAM_STICKY (training loops re-bind every step): only the first bind reads the pack's verdict back.  Parameters
    that stop fitting the expansion behind a stale "fits" verdict are caught on the device -- every wave of K1g takes the VALU
    distances, K2g hands over to K2h -- and the results stay right."""
    import torch
    rng = np.random.default_rng(21)
    cfg, P = configs.model_card(2) | {"num_kernels": 256}, configs.synth_params(2)
    P = {"params": {"rbf_list": {k: v[:, :256].copy() for k, v in P["params"]["rbf_list"].items()},
                    "linear": {"kernel": P["params"]["linear"]["kernel"][:256].copy(), "bias": P["params"]["linear"]["bias"].copy()}}}
    net = WCRBFNet.from_config(cfg)
    B = 16384 + 5
    x = configs.synth_queries(2, B=B)
    g = rng.normal(size=(B, 10)).astype(np.float32)
    net.apply(P, x)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram<")
    net.set_options(gram_sticky=1)
    P2 = {"params": {"rbf_list": {"centers": P["params"]["rbf_list"]["centers"].copy(),
                                  "log_sigs": np.full_like(P["params"]["rbf_list"]["log_sigs"], -7.0)},      # widths of 1e-3: outside the budget
                     "linear": P["params"]["linear"]}}
    P2["params"]["rbf_list"]["log_sigs"][0, :32] = 0.5                     # a few centres the queries still see
    out = net.apply(P2, x)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram<")       # the stale verdict chose the kernel ...
    p64 = orc.cast_params(P2, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
    k1, _ = _run(net, P2, x, kernel=_lib.FWD_K1)
    err, err_k1 = (np.abs(out - ref) / scale).max(), (np.abs(k1 - ref) / scale).max()
    assert err <= max(3e-6, 2.0 * err_k1), (err, err_k1)                    # ... and the device-side test kept the result right
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    grads = net.vjp(P2, xt, gt)["params"]
    gref = orc.wcrbfnet_vjp(cfg, p64, x.astype(np.float64), g.astype(np.float64))["params"]
    for grp, name in LEAVES:
        gr = np.asarray(gref[grp][name])
        assert np.abs(grads[grp][name].cpu().numpy() - gr).max() <= 2e-5 * (np.abs(gr).max() + 1e-30), (grp, name)
    net.set_options(gram_sticky=0)
    net.bind(P)                                            # different parameters: re-packed, verdict read back again
    net.apply(P2, x)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16mfma<")       # without the option the fresh verdict moves the net to K1h


def test_training_steps_on_the_matrix_core_kernels_follow_the_float32_ones(gpu):
    """Six on-device training steps (scripts/train_nmpc.py:258-300) at a batch the automatic choice gives to K1g + K2g, against the same
    steps forced onto K1h + K2h: same losses (1e-5), finite parameters, and the VJP really ran on K2g."""
    import torch
    from irbfn_amd import train
    card = dict(configs.model_card(3), num_kernels=2048)            # 16448 x 2048 = 3.4e7 pairs: above K2g's threshold of 2.5e7
    P = configs.synth_params(3)
    P = {"params": {"rbf_list": {k: v[:, :2048] for k, v in P["params"]["rbf_list"].items()},
                    "linear": {"kernel": P["params"]["linear"]["kernel"][:2048], "bias": P["params"]["linear"]["bias"]}}}
    B = 16384 + 64
    x = torch.from_numpy(configs.synth_queries(3, B=B)).cuda()
    y = torch.from_numpy(configs.synth_cotangent(3, B=B)).cuda()
    runs = {}
    for name, fk, vk in (("h", _lib.FWD_K1H, _lib.VJP_K2H), ("g", _lib.FWD_AUTO, _lib.VJP_AUTO)):
        net = WCRBFNet.from_config(card)
        net.set_options(fwd_kernel=fk, vjp_kernel=vk)
        st = train.TrainState.create(net, P, lr=1e-3, max_grad_norm=1.0)
        losses = []
        for _ in range(6):
            st, loss = train.train_step_oneint(st, x, y, configs.DYN_PARAMS)
            losses.append(float(loss))
        runs[name] = (losses, st.flat.clone(), net.last_launch()["kernel"])
    assert runs["g"][2].startswith("rbf_vjp_f16gram<"), runs["g"][2]
    assert np.isfinite(runs["g"][0]).all() and bool(torch.isfinite(runs["g"][1]).all())
    assert np.allclose(runs["g"][0], runs["h"][0], rtol=1e-5)
    assert float((runs["g"][1] - runs["h"][1]).abs().max()) <= 1e-4 * float(runs["h"][1].abs().max())
