"""GPU parity of K1h / K2h, the forward and the parameter VJP with their GEMM-shaped pieces on the f16 matrix
cores at float32 accuracy (irbfn_amd/csrc/rbf_forward_f16.hip, rbf_vjp_f16.hip, f16_split.h), selected through
the descriptor options (irbfn_net_set_option): against the float64 oracle -- scaled by sum_k |phi_k W_k| (the
natural error scale of the reduction) and, on ill-conditioned weight columns / cotangent batches built without
cancellation, at the north-star's 1e-5 relative to |ref| -- and against the all-float32 kernels K1 / K2."""
import numpy as np
import pytest

from conftest import load_ckpt_fixture
from irbfn_amd import _lib, configs
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu


class _opts:
    """Sets descriptor options for a block and restores the defaults afterwards."""
    DEFAULTS = {"fwd_kernel": 0, "fwd_f16_terms": 3, "fwd_f16_s": 0, "fwd_f16_qg": 0, "vjp_kernel": 0, "vjp_f16_ct": 0}

    def __init__(self, net, **kv):
        self.net, self.kv = net, kv

    def __enter__(self):
        self.net.set_options(**self.kv)

    def __exit__(self, *exc):
        self.net.set_options(**{k: self.DEFAULTS[k] for k in self.kv})


def _terms_scale(cfg, p64, x):
    """sum_k |gamma phi_k W_k| + |bias| per (query, output): float64."""
    pa = {"params": {"rbf_list": p64["params"]["rbf_list"],
                     "linear": {"kernel": np.abs(p64["params"]["linear"]["kernel"]), "bias": np.abs(p64["params"]["linear"]["bias"])}}}
    return orc.wcrbfnet_apply(cfg, pa, x)


def _run(net, params, x, **opts):
    with _opts(net, fwd_kernel=_lib.FWD_K1H, **opts):
        got = net.apply(params, x)
        name = net.last_launch()["kernel"]
    return got, name


def test_f16mfma_cfg2_accuracy_and_agreement(gpu):
    cfg = configs.model_card(2)
    params = configs.synth_params(2)
    net = WCRBFNet.from_config(cfg)
    B = 4096 + 37                                    # ragged tail
    x = configs.synth_queries(2, B=B)
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16mfma<D=7,BC=0,TERMS=3"), name
    p64 = orc.cast_params(params, np.float64)
    x64 = x.astype(np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x64)
    scale = _terms_scale(cfg, p64, x64)
    err = np.abs(got - ref) / scale
    with _opts(net, fwd_kernel=_lib.FWD_K1):
        k1 = net.apply(params, x)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_qlane")
    err_k1 = np.abs(k1 - ref) / scale
    print(f"K1h max/mean err {err.max():.2e}/{err.mean():.2e}   K1 max/mean err {err_k1.max():.2e}/{err_k1.mean():.2e}")
    # float32-equivalent: no worse than the float32 FMA-chain kernel on the same inputs
    assert err.max() <= 3e-6 and err.max() <= 2.0 * err_k1.max() and err.mean() <= 1.5 * err_k1.mean(), (err.max(), err_k1.max())
    assert (np.abs(got - k1) / scale).max() <= 1e-5
    # every (S, QG) geometry gives the same answer up to the slice summation order
    for S, QG in ((1, 8), (2, 4), (4, 2), (8, 1), (4, 1), (1, 1)):
        g2, nm = _run(net, params, x, fwd_f16_s=S, fwd_f16_qg=QG)
        assert f"S={S},QG={QG}" in nm
        assert (np.abs(g2 - ref) / scale).max() <= 3e-6, (S, QG)
    # reduced-precision single-product variant (reporting only): plain f16 operands
    g1, nm = _run(net, params, x, fwd_f16_terms=1)
    assert "TERMS=1" in nm
    e1 = (np.abs(g1 - ref) / scale).max()
    assert 3e-6 < e1 <= 2e-3, e1
    gb, nm = _run(net, params, x, fwd_f16_terms=2)                     # plain bf16 operands (BASELINE config 5's "bf16")
    assert "BF16" in nm
    eb = (np.abs(gb - ref) / scale).max()
    assert e1 < eb <= 2e-2, (e1, eb)


@pytest.mark.parametrize("run", ["dnmpc_1regions_newdata_oldintloss_nomirror_highk",
                                 "dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq"])
@pytest.mark.parametrize("B", [100, 2500])
def test_f16mfma_on_reference_checkpoints(gpu, run, B):
    """Trained single-region nets of the reference (gaussian O = 10, inverse_quadratic O = 2; K = 1000 is not
    a multiple of the 32-centre chunk; bounds make gamma != 1 near the edges)."""
    cfg, params, x0, out64, *_ = load_ckpt_fixture(run)
    net = WCRBFNet.from_config(cfg)
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)])
    hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = np.random.default_rng(B).uniform(lo - 0.02, hi + 0.02, size=(B, cfg["in_features"])).astype(np.float32)
    x[:64] = x0.astype(np.float32)
    p32 = orc.cast_params(params, np.float32)
    got, name = _run(net, p32, x)
    assert name.startswith("rbf_fwd_f16mfma"), name
    p64 = orc.cast_params(p32, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
    assert (np.abs(got - ref) / scale).max() <= 2e-6
    gam = orc.region_activation(x.astype(np.float64), cfg["num_regions"], len(cfg["activation_idx"]), cfg["lower_bounds"],
                                cfg["upper_bounds"], cfg["delta"], cfg["dimension_ranges"])
    assert gam.min() < 0.9 and gam.max() > 0.99      # the gate is exercised


@pytest.mark.parametrize("D,K,O,basis", [(3, 256, 5, "gaussian"), (4, 96, 16, "inverse_multiquadric"),
                                         (8, 200, 2, "inverse_quadratic"), (7, 33, 1, "gaussian_wide"),
                                         (5, 64, 7, "inverse_multiquadric")])
def test_f16mfma_shapes_and_bases(gpu, D, K, O, basis):
    rng = np.random.default_rng(D * 100 + K)
    lo, hi = -np.ones(D) * 2, np.ones(D) * 3
    cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": basis, "num_regions": 1,
           "lower_bounds": [[float(v)] for v in lo], "upper_bounds": [[float(v)] for v in hi],
           "dimension_ranges": [[0] * D], "activation_idx": list(range(D)), "delta": [20.0] * D}
    params = {"params": {"rbf_list": {"centers": rng.uniform(lo - 1, hi + 1, size=(1, K, D)).astype(np.float32),
                                      "log_sigs": rng.uniform(-0.5, 1.5, size=(1, K)).astype(np.float32)},
                         "linear": {"kernel": (rng.normal(size=(K, O)) * rng.choice([1e-3, 1.0, 300.0], size=(1, O))).astype(np.float32),
                                    "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    B = 1000
    x = rng.uniform(lo, hi, size=(B, D)).astype(np.float32)
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16mfma"), name
    p64 = orc.cast_params(params, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
    assert (np.abs(got - ref) / scale).max() <= 2e-6


def test_f16mfma_not_used_where_ineligible(gpu):
    """Multi-region nets are not eligible: forcing K1h is refused (IRBFN_ERR_UNSUPPORTED -> ValueError), the
    automatic choice is a float32 kernel of the multi-region family (the region-sparse K1r for this sparse gate)."""
    cfg, params, x, *_ = load_ckpt_fixture("dnmpc_128regions")
    net = WCRBFNet.from_config(cfg)
    xs = np.repeat(x.astype(np.float32), 4, axis=0)
    with pytest.raises(ValueError):
        _run(net, orc.cast_params(params, np.float32), xs)
    net.apply(orc.cast_params(params, np.float32), xs)
    assert net.last_launch()["kernel"].startswith(("rbf_fwd_sparse", "rbf_fwd_qlane"))
    # NaN queries propagate (IEEE), other rows are unaffected
    cfg2, p2 = configs.model_card(2), configs.synth_params(2)
    net2 = WCRBFNet.from_config(cfg2)
    x2 = configs.synth_queries(2, B=256)
    clean, _ = _run(net2, p2, x2)
    x2n = x2.copy()
    x2n[5, 3] = np.nan
    dirty, _ = _run(net2, p2, x2n)
    assert np.isnan(dirty[5]).all()
    keep = np.ones(256, bool); keep[5] = False
    np.testing.assert_array_equal(dirty[keep], clean[keep])


@pytest.mark.parametrize("O,SW", [(100, 2), (100, 1), (100, 4), (20, 2), (64, 1), (128, 4)])
def test_f16mfma_wide_outputs(gpu, O, SW):
    """16 < O <= 128 (50-step control sequences, O = 100): block-shared W stream, NT column tiles."""
    rng = np.random.default_rng(O + SW)
    D, K = 7, 300
    cfg = dict(configs.model_card(4), num_kernels=K, out_features=O)
    params = {"params": {"rbf_list": {"centers": rng.uniform(-1, 8, size=(1, K, D)).astype(np.float32),
                                      "log_sigs": rng.uniform(0.0, 2.0, size=(1, K)).astype(np.float32)},
                         "linear": {"kernel": rng.normal(0, 0.3, size=(K, O)).astype(np.float32),
                                    "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    B = 777
    x = configs.synth_queries(4, B=B)
    got, name = _run(net, params, x, fwd_f16_s=SW)
    assert name.startswith("rbf_fwd_f16mfma_wide") and f"SW={SW}" in name, name
    p64 = orc.cast_params(params, np.float64)
    ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    scale = _terms_scale(cfg, p64, x.astype(np.float64)) + 1e-30
    assert got.shape == (B, O)
    with _opts(net, fwd_kernel=_lib.FWD_K1M):
        other = net.apply(params, x)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_mfma")
    err, err_other = (np.abs(got - ref) / scale).max(), (np.abs(other - ref) / scale).max()
    # float32-equivalent: the error is that of the shared float32 distance / basis arithmetic
    assert err <= max(2e-6, 2.0 * err_other), (err, err_other)
    assert (np.abs(got - other) / scale).max() <= 1e-5


LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


@pytest.mark.parametrize("CT", [2, 4])
@pytest.mark.parametrize("case", ["cfg3", "iq_ckpt", "imq_d3"])
def test_vjp_f16mfma_matches_valu_kernel_and_oracle(gpu, case, CT):
    """K2h (rbf_vjp_f16.hip): hbar and dW on the f16 matrix cores with hi/lo operand splits -- against K2 (all
    float32 VALU) on the same inputs and against the float64 restatement of the reference's parameter VJP."""
    import torch
    rng = np.random.default_rng(11)
    if case == "cfg3":
        cfg = dict(configs.model_card(3), num_kernels=500)                 # not a multiple of 32 / 64
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
    else:
        D, K, O = 3, 96, 5
        cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": "inverse_multiquadric", "num_regions": 1,
               "lower_bounds": [[-2.0]] * D, "upper_bounds": [[3.0]] * D, "dimension_ranges": [[0] * D],
               "activation_idx": list(range(D)), "delta": [20.0] * D}
        P = {"params": {"rbf_list": {"centers": rng.uniform(-3, 4, size=(1, K, D)).astype(np.float32),
                                     "log_sigs": rng.uniform(-0.5, 1.0, size=(1, K)).astype(np.float32)},
                        "linear": {"kernel": (rng.normal(size=(K, O)) * np.array([1e-2, 1, 1, 50, 1])).astype(np.float32),
                                   "bias": rng.normal(size=(O,)).astype(np.float32)}}}
        B = 2048 + 1
        x = rng.uniform(-2.2, 3.2, size=(B, D)).astype(np.float32)
        g = rng.normal(size=(B, O)).astype(np.float32)
        g[7] = 0.0
    net = WCRBFNet.from_config(cfg)
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    with _opts(net, vjp_kernel=_lib.VJP_K2H, vjp_f16_ct=CT):
        a = net.vjp(P, xt, gt)["params"]
        a2 = net.vjp(P, xt, gt)["params"]
    with _opts(net, vjp_kernel=_lib.VJP_K2):
        b = net.vjp(P, xt, gt)["params"]
    ref = orc.wcrbfnet_vjp(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), g.astype(np.float64))["params"]
    for grp, name in LEAVES:
        ga, gb, gr = a[grp][name].cpu().numpy(), b[grp][name].cpu().numpy(), np.asarray(ref[grp][name])
        assert torch.equal(a[grp][name], a2[grp][name])                    # deterministic
        scale = np.abs(gr).max() + 1e-30
        ea, eb = np.abs(ga - gr).max() / scale, np.abs(gb - gr).max() / scale
        assert ea <= max(2e-5, 2.0 * eb), (case, grp, name, ea, eb)


# ---- ill-conditioned operands: the hi/lo pairs must not depend on how a factor compares with the maximum its
# ---- power-of-two scale was taken from (f16_split.h).  Built WITHOUT cancellation (non-negative weights and
# ---- cotangents), so that the north-star's "1e-5 relative to |ref|" is a meaningful bound per output.
def _cond_net(K=512, O=10, basis="gaussian", seed=0):
    rng = np.random.default_rng(seed)
    D = 7
    lo, hi = np.zeros(D), np.full(D, 4.0)
    cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": basis, "num_regions": 1,
           "lower_bounds": [[-100.0]] * D, "upper_bounds": [[100.0]] * D, "dimension_ranges": [[0] * D],
           "activation_idx": list(range(D)), "delta": [10.0] * D}          # gate == 1 on the queries
    centers = rng.uniform(lo, hi, size=(1, K, D)).astype(np.float32)
    log_sigs = rng.uniform(-0.7, 0.3, size=(1, K)).astype(np.float32)
    return rng, cfg, centers, log_sigs


@pytest.mark.parametrize("basis", ["gaussian", "inverse_quadratic", "inverse_multiquadric"])
@pytest.mark.parametrize("case", ["outlier_far_centre", "six_decades", "queries_on_small_weight_centres", "columns_1e-6_to_1e6"])
def test_forward_f16_ill_conditioned_columns(gpu, case, basis):
    K, O = 512, 10
    rng, cfg, centers, log_sigs = _cond_net(K, O, basis, seed=len(case))
    B = 2048 + 5
    x = rng.uniform(0.0, 4.0, size=(B, 7)).astype(np.float32)
    W = np.abs(rng.normal(size=(K, O))) + 0.05
    if case == "outlier_far_centre":
        centers[0, 3] = 60.0                               # far from every query: phi ~ 0 (gaussian) / ~1e-4 (IQ)
        log_sigs[0, 3] = -0.5
        W[3, :] = 1.0e4                                    # sets every column's scale 1e4 above the bulk
    elif case == "six_decades":
        W = 10.0 ** rng.uniform(-3, 3, size=(K, O))
    elif case == "queries_on_small_weight_centres":
        W = 10.0 ** rng.uniform(-3, 3, size=(K, O))
        small = np.argsort(W.max(axis=1))[:64]             # the 64 centres whose whole weight row is smallest
        x[:1024] = centers[0, small[rng.integers(0, 64, size=1024)]] + rng.normal(0, 0.02, size=(1024, 7)).astype(np.float32)
        log_sigs[0, small] = -1.5                          # narrow: those queries see (almost) only their own centre
    else:
        W = W * 10.0 ** np.linspace(-6, 6, O)[None, :]     # per-column scales (the part the old split handled too)
    params = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs},
                         "linear": {"kernel": W.astype(np.float32), "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16mfma<"), name
    with _opts(net, fwd_kernel=_lib.FWD_K1):
        k1 = net.apply(params, x)
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x.astype(np.float64))
    assert (ref > 0).all()
    rel, rel_k1 = np.abs(got - ref) / ref, np.abs(k1 - ref) / ref
    print(f"{case}/{basis}: K1h max rel {rel.max():.2e} (K1 {rel_k1.max():.2e}), median {np.median(rel):.2e}")
    assert rel.max() <= 1e-5, (case, basis, rel.max(), rel_k1.max())        # north-star: 1e-5 relative to |ref|
    assert rel.max() <= max(3e-6, 2.5 * rel_k1.max())                        # and float32-grade: as good as K1


@pytest.mark.parametrize("O,case", [(100, "outlier_far_centre"), (100, "six_decades"), (40, "six_decades")])
def test_forward_f16_wide_ill_conditioned_columns(gpu, O, case):
    K = 300
    rng, cfg, centers, log_sigs = _cond_net(K, O, "gaussian", seed=O)
    B = 1024 + 3
    x = rng.uniform(0.0, 4.0, size=(B, 7)).astype(np.float32)
    W = np.abs(rng.normal(size=(K, O))) + 0.05
    if case == "outlier_far_centre":
        centers[0, 5] = 60.0
        W[5, :] = 1.0e4
    else:
        W = 10.0 ** rng.uniform(-3, 3, size=(K, O))
    params = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs},
                         "linear": {"kernel": W.astype(np.float32), "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    got, name = _run(net, params, x)
    assert name.startswith("rbf_fwd_f16mfma_wide"), name
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x.astype(np.float64))
    rel = np.abs(got - ref) / ref
    print(f"wide {case}/O={O}: max rel {rel.max():.2e}")
    assert rel.max() <= 1e-5, (case, O, rel.max())


@pytest.mark.parametrize("case", ["outlier_cotangent_row", "six_decade_rows", "outlier_weight"])
def test_vjp_f16_ill_conditioned_cotangents(gpu, case):
    """K2h scales the cotangent batch by ONE power of two (batch max |g|) and the weights per column: bulk rows
    far below an outlier row (and weights far below their column's outlier) must keep float32 accuracy."""
    import torch
    K, O = 256, 10
    rng, cfg, centers, log_sigs = _cond_net(K, O, "gaussian", seed=7)
    B = 4096 + 9
    x = rng.uniform(0.0, 4.0, size=(B, 7)).astype(np.float32)
    W = (np.abs(rng.normal(size=(K, O))) + 0.05).astype(np.float32)
    g = (np.abs(rng.normal(size=(B, O))) + 0.05).astype(np.float32)
    if case == "outlier_cotangent_row":
        x[0] = 50.0                                        # the outlier query sees no centre: only the bulk rows count
        g[0] = 1.0e4
    elif case == "six_decade_rows":
        g *= (10.0 ** rng.uniform(-3, 3, size=(B, 1))).astype(np.float32)
    else:
        centers[0, 9] = 60.0
        W[9, :] = 1.0e4
    P = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs},
                    "linear": {"kernel": W, "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    with _opts(net, vjp_kernel=_lib.VJP_K2H):
        a = net.vjp(P, xt, gt)["params"]
    with _opts(net, vjp_kernel=_lib.VJP_K2):
        b = net.vjp(P, xt, gt)["params"]
    ref = orc.wcrbfnet_vjp(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), g.astype(np.float64))["params"]
    live = np.ones(K, bool)
    if case == "outlier_weight":
        live[9] = False                                    # the far centre itself has ~zero gradients
    # dW and d log_sigs are sums of same-signed terms here: 1e-5 relative per entry (bias is an exact column sum)
    for grp, name in (("linear", "kernel"), ("rbf_list", "log_sigs")):
        ga, gb, gr = a[grp][name].cpu().numpy(), b[grp][name].cpu().numpy(), np.asarray(ref[grp][name])
        ga, gb, gr = ga.reshape(K, -1)[live], gb.reshape(K, -1)[live], gr.reshape(K, -1)[live]
        ra, rb = np.abs(ga - gr) / np.abs(gr), np.abs(gb - gr) / np.abs(gr)
        print(f"{case} {name}: K2h max rel {ra.max():.2e}  K2 {rb.max():.2e}")
        assert ra.max() <= 1e-5, (case, name, ra.max(), rb.max())
    # d centers has mixed-sign terms: relative to the centre's largest component, and no worse than 3x K2
    ga, gb, gr = (v.reshape(K, 7)[live] for v in (a["rbf_list"]["centers"].cpu().numpy(), b["rbf_list"]["centers"].cpu().numpy(),
                                                   np.asarray(ref["rbf_list"]["centers"])))
    rowmax = np.abs(gr).max(axis=1, keepdims=True)
    ra, rb = np.abs(ga - gr) / rowmax, np.abs(gb - gr) / rowmax
    print(f"{case} centers: K2h max {ra.max():.2e}  K2 {rb.max():.2e}")
    assert ra.max() <= max(1e-5, 3.0 * rb.max()), (case, ra.max(), rb.max())


def test_forward_f16_far_queries_error_floor(gpu):
    """Queries far from every centre (all basis values tiny): the documented floor of the f16 pairs
    (f16_split.h) -- float32-grade terms down to phi = 2^-28, absolute error <= 2^-38 per unit weight below."""
    K, O, B = 512, 10, 1024
    rng, cfg, centers, log_sigs = _cond_net(K, O, "gaussian", seed=3)
    centers[:] = rng.uniform(0, 1, size=centers.shape)
    log_sigs[:] = 0.0
    W = (np.abs(rng.normal(size=(K, O))) + 0.05).astype(np.float32)
    params = {"params": {"rbf_list": {"centers": centers, "log_sigs": log_sigs},
                         "linear": {"kernel": W, "bias": np.zeros(O, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    p64 = orc.cast_params(params, np.float64)
    for shift, rel_tol in ((0.0, 2e-6), (2.0, 2e-6), (3.0, 3e-6), (4.0, None), (5.0, None)):
        x = rng.uniform(0, 1, size=(B, 7)).astype(np.float32)
        x[:, 0] += shift
        got, _ = _run(net, params, x)
        ref = orc.wcrbfnet_apply(cfg, p64, x.astype(np.float64))
        err = np.abs(got - ref)
        if rel_tol is not None:
            assert (err / ref).max() <= rel_tol, (shift, (err / ref).max())
        assert (err <= 3e-6 * ref + 2.0 ** -38 * W.sum(axis=0)[None, :]).all(), shift
