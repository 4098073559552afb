"""Region-sparse evaluation of the multi-region nets (rbf_sparse.hip, K1r / K2r) against the dense gated kernels and
the float64 oracle.  The reference evaluates every region for every query (src/irbfn_mpc/model.py:187-193); a region
whose gamma is exactly 0 in float32 contributes nothing, so visiting only the regions with gamma != 0 must reproduce
the dense result up to the ORDER of the float32 sum.

Bound used below for sparse vs dense: both are float32 sums of the same <= R*K products gamma*phi*W in a different
order, |a - b| <= 2 n eps sum_k |h_k W_ko| with n eps = 1280 * 6e-8 -> 2e-6 * sum|h W| (+ 1e-7 * |ref| for the bias add).
"""
import numpy as np
import pytest

from conftest import load_ckpt_fixture
from irbfn_amd import _lib, configs, planner
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu
RUNS = ["dnmpc_128regions", "dnmpc_12regions_frenet_l1_bigdata"]


def _queries(cfg, B, seed=0, border=False):
    """In-range queries; border=True puts every split coordinate ON (or within 0.02 of) a range boundary, so that gamma
    is strictly between 0 and 1 in several regions at once."""
    rng = np.random.default_rng(seed)
    ns, D = len(cfg["activation_idx"]), cfg["in_features"]
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)])
    hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = np.hstack([rng.uniform(lo, hi, size=(B, ns)), rng.normal(size=(B, D - ns)) * 0.1])
    if border:
        for d in range(ns):
            edges = np.array(sorted(set(cfg["lower_bounds"][d]) | set(cfg["upper_bounds"][d])))
            x[:, d] = edges[rng.integers(0, len(edges), B)] + rng.choice([0.0, 0.02, -0.02, 0.3 / cfg["delta"][d]], B)
    return x.astype(np.float32)


def _cancel(cfg, P, x):
    _, h, gamma = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), return_aux=True)
    return np.abs(h) @ np.abs(np.asarray(P["params"]["linear"]["kernel"], np.float64)), gamma


@pytest.mark.parametrize("run", RUNS)
@pytest.mark.parametrize("B", [65, 511, 513, 3000])
def test_sparse_forward_matches_dense_and_oracle(gpu, run, B):
    cfg, P, *_ = load_ckpt_fixture(run)
    for border in (False, True):
        x = _queries(cfg, B, seed=B, border=border)
        ref = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64))
        cancel, gamma = _cancel(cfg, P, x)
        if border:                                  # the case is what it claims: several regions strictly inside (0, 1)
            assert (((gamma > 1e-3) & (gamma < 0.999)).sum(axis=1) >= 2).mean() > 0.25
        net = WCRBFNet.from_config(cfg)
        net.set_options(fwd_kernel=_lib.FWD_K1R)
        sp = net.apply(P, x)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_sparse<")
        net.set_options(fwd_kernel=_lib.FWD_K1)
        dn = net.apply(P, x)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_qlane<")
        assert (np.abs(sp - dn) <= 2e-6 * cancel + 1e-7 * np.abs(ref)).all(), (run, B, border, np.abs(sp - dn).max())
        assert (np.abs(sp - ref) <= 1e-5 * np.abs(ref) + 3e-6 * cancel).all(), (run, B, border)
        net.set_options(fwd_kernel=_lib.FWD_AUTO)   # automatic dispatch: sparse where the gate is clearly sparse (3.6 of 128
        au = net.apply(P, x)                        # regions live per query), dense where half the regions are (5.7 of 12)
        if run == "dnmpc_128regions":
            assert net.last_launch()["kernel"].startswith("rbf_fwd_sparse<") and np.array_equal(au, sp)
        else:
            assert net.last_launch()["kernel"].startswith("rbf_fwd_qlane<") and np.array_equal(au, dn)


def test_sparse_forward_is_independent_of_the_batch_around_a_query(gpu):
    """A query's result does not depend on which lane served it or what else is in the batch: bit-identical when the
    batch is permuted and when the query is evaluated with a different neighbourhood."""
    cfg, P, *_ = load_ckpt_fixture("dnmpc_128regions")
    x = _queries(cfg, 2000, seed=5)
    net = WCRBFNet.from_config(cfg)
    net.set_options(fwd_kernel=_lib.FWD_K1R)
    a = net.apply(P, x)
    perm = np.random.default_rng(1).permutation(len(x))
    b = net.apply(P, x[perm])
    assert np.array_equal(a[perm], b)
    c = net.apply(P, x[:700])
    assert np.array_equal(a[:700], c)


def test_sparse_forward_nan_inf_and_short_dimension_ranges(gpu):
    cfg, P, *_ = load_ckpt_fixture("dnmpc_128regions")
    x = _queries(cfg, 300, seed=9)
    x[3, 1] = np.nan
    x[5, 0] = np.inf
    x[7, 2] = -np.inf
    x[9, :] = np.nan                                  # every factor NaN: more "active" regions than the list holds
    for c in (cfg, dict(cfg, dimension_ranges=cfg["dimension_ranges"][:100])):      # App. B-2: 28 regions stay 0
        with np.errstate(all="ignore"):
            ref = orc.wcrbfnet_apply(c, orc.cast_params(P, np.float64), x.astype(np.float64))
        net = WCRBFNet.from_config(c)
        net.set_options(fwd_kernel=_lib.FWD_K1R)
        sp = net.apply(P, x)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_sparse<")
        assert np.isnan(sp[[3, 9]]).all() and np.isnan(ref[[3, 9]]).all()
        ok = np.ones(len(x), bool); ok[[3, 9]] = False
        assert np.isfinite(sp[ok]).all()
        cancel, _ = _cancel(c, P, np.nan_to_num(x, nan=0.0, posinf=1e30, neginf=-1e30))
        assert (np.abs(sp[ok] - ref[ok]) <= 1e-5 * np.abs(ref[ok]) + 3e-6 * cancel[ok]).all()
        # x = +-inf: the gate closes every region (one factor is exactly 0) -> bias, as the oracle
        assert np.allclose(sp[[5, 7]], ref[[5, 7]], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("run,mode", [("dnmpc_128regions", _lib.ROLLOUT_ST_SELECT), ("dnmpc_128regions", _lib.ROLLOUT_ST_KS),
                                      ("dnmpc_128regions", _lib.ROLLOUT_FULLINT)])
def test_sparse_tick_equals_forward_then_rollout(gpu, run, mode):
    """irbfn_plan_tick on a multi-region planner net: forward + sign flip of the mirrored rows + 5-step roll-out in ONE
    launch of the sparse kernel == sparse forward -> un-mirror -> stand-alone roll-out, bit for bit (shared step
    functions), with and without mirror flags, ragged batch."""
    import torch
    from irbfn_amd import dynamics
    cfg, P, *_ = load_ckpt_fixture(run)
    net = WCRBFNet.from_config(cfg)
    B = 1300
    x = torch.from_numpy(_queries(cfg, B, seed=3)).cuda()
    rng = np.random.default_rng(4)
    if mode == _lib.ROLLOUT_FULLINT:
        s0 = torch.from_numpy(rng.uniform(0, 7, size=(B, 1)).astype(np.float32)).cuda()
    else:
        s0 = torch.from_numpy(configs.initial_state_from_query(x.cpu().numpy())).cuda()
    mirror = torch.from_numpy((rng.random(B) < 0.4).astype(np.int32)).cuda()
    for m in (None, mirror):
        net.set_options(fwd_kernel=_lib.FWD_K1R)
        ctrl, states = planner.plan_tick(net, P, x, m, s0, configs.DYN_PARAMS, mode=mode)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_sparse<") and "ROLL=1" in net.last_launch()["kernel"]
        u = net.apply(P, x).clone()
        if m is not None:
            u[m.bool(), 5:] *= -1.0
        assert torch.equal(ctrl, u)
        if mode == _lib.ROLLOUT_FULLINT:
            ref_states = dynamics.rollout_fullint(s0.reshape(-1), u)
        elif mode == _lib.ROLLOUT_ST_KS:
            ref_states = dynamics.integrate_st_ks_mult(torch.cat([s0, u], dim=1), configs.DYN_PARAMS)
        else:
            ref_states = dynamics.integrate_st_mult(torch.cat([s0, u], dim=1), configs.DYN_PARAMS)
        assert torch.equal(states, ref_states.reshape(states.shape))


# ---------------------------------------------------------------------------------------------------------------
# K2r: region-sparse parameter VJP
# ---------------------------------------------------------------------------------------------------------------
LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


@pytest.mark.parametrize("run", RUNS)
@pytest.mark.parametrize("B", [1, 63, 257, 5000])
def test_sparse_vjp_matches_dense_and_oracle(gpu, run, B):
    """Four gradient leaves of the pair-list VJP against the dense gated K2 (same terms, other order) and the float64 C
    restatement of the reference's backward (oracle_wcrbf_vjp); ragged batches; border queries (gamma strictly inside
    (0, 1) in several regions); bitwise reproducible (no position comes from an atomic)."""
    import torch
    from oracle import c_oracle as co
    cfg, P, *_ = load_ckpt_fixture(run)
    P32 = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
    rng = np.random.default_rng(B)
    for border in (False, True):
        x = _queries(cfg, B, seed=B + 1, border=border)
        g = rng.normal(size=(B, cfg["out_features"])).astype(np.float32)
        xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
        ref = co.wcrbf_vjp(cfg, P32, x, g, np.float64)["params"]
        net = WCRBFNet.from_config(cfg)
        net.set_options(vjp_kernel=_lib.VJP_K2R)
        sp = net.vjp(P32, xt, gt)["params"]
        sp2 = net.vjp(P32, xt, gt)["params"]
        net.set_options(vjp_kernel=_lib.VJP_K2)
        dn = net.vjp(P32, xt, gt)["params"]
        for grp, name in LEAVES:
            a, b, d = sp[grp][name], sp2[grp][name], dn[grp][name]
            assert torch.equal(a, b), (run, B, name)                          # deterministic
            r = ref[grp][name]
            scale = np.abs(r).max() + 1e-30
            ea = np.abs(a.cpu().numpy() - r).max() / scale
            ed = np.abs(d.cpu().numpy() - r).max() / scale
            assert ea <= 5e-5, (run, B, border, name, ea, ed)                 # the bound the dense kernels are held to
            assert np.abs(a.cpu().numpy() - d.cpu().numpy()).max() <= 2e-5 * scale, (run, B, border, name)


def test_sparse_vjp_nan_query_and_forced_kernel_errors(gpu):
    import torch
    cfg, P, *_ = load_ckpt_fixture("dnmpc_128regions")
    P32 = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
    x = _queries(cfg, 700, seed=2)
    x[11, :] = np.nan                                   # more live factors than any finite query: the list is capped
    g = np.random.default_rng(0).normal(size=(700, 10)).astype(np.float32)
    net = WCRBFNet.from_config(cfg)
    net.set_options(vjp_kernel=_lib.VJP_K2R)
    out = net.vjp(P32, x, g)["params"]
    assert np.isnan(out["linear"]["kernel"]).any()      # NaN reaches the gradients, as in the dense kernel / jax
    # a one-region net cannot take the sparse kernels: the forced option fails loudly instead of falling back
    net1 = WCRBFNet.from_config(configs.model_card(1))
    net1.set_options(vjp_kernel=_lib.VJP_K2R)
    with pytest.raises(Exception):
        net1.vjp(configs.synth_params(1), configs.synth_queries(1, B=100), configs.synth_cotangent(1, B=100))
    net1.set_options(fwd_kernel=_lib.FWD_K1R)
    with pytest.raises(Exception):
        net1.apply(configs.synth_params(1), configs.synth_queries(1, B=100))


def test_train_step_on_the_128_region_net_uses_the_sparse_kernels(gpu):
    """train_step_fullint (scripts/train_nmpc.py:303-421) at the reference's batch size on its 128-region planner:
    automatic dispatch = K1r forward + K2r VJP; two steps equal the dense kernels' two steps to float32 reordering."""
    import torch
    from irbfn_amd import train
    cfg, P, *_ = load_ckpt_fixture("dnmpc_128regions")
    P32 = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
    B = 20000
    rng = np.random.default_rng(3)
    x = torch.from_numpy(_queries(cfg, B, seed=8)).cuda()
    y = torch.from_numpy(np.hstack([rng.normal(size=(B, 5)) * 2, rng.normal(size=(B, 5)) * 0.5]).astype(np.float32)).cuda()
    res = {}
    for name, kf, kv in (("sparse", _lib.FWD_AUTO, _lib.VJP_AUTO), ("dense", _lib.FWD_K1, _lib.VJP_K2)):
        net = WCRBFNet.from_config(cfg)
        net.set_options(fwd_kernel=kf, vjp_kernel=kv)
        st = train.TrainState.create(net, P32, lr=1e-3, max_grad_norm=1.0)
        losses = []
        for _ in range(2):
            st, loss = train.train_step_fullint(st, x, y)
            losses.append(float(loss))
        if name == "sparse":
            assert net.last_launch()["kernel"].startswith("rbf_fwd_sparse<")
        res[name] = (losses, st.flat.clone())
    assert np.allclose(res["sparse"][0], res["dense"][0], rtol=2e-5)
    assert (res["sparse"][1] - res["dense"][1]).abs().max() <= 1e-4 * res["dense"][1].abs().max()


def test_sparse_forward_large_batch_gathers_the_centre_table_from_global_memory(gpu):
    """More workgroups than CUs: the kernel variant that leaves the centre table in global memory (three workgroups per CU)
    is taken; same arithmetic in the same order as the LDS variant -> bit-identical rows, and the tick likewise."""
    import torch
    cfg, P, *_ = load_ckpt_fixture("dnmpc_128regions")
    B = 70001
    x = torch.from_numpy(_queries(cfg, B, seed=12)).cuda()
    net = WCRBFNet.from_config(cfg)
    net.set_options(fwd_kernel=_lib.FWD_K1R)
    big = net.apply(P, x)
    assert "GC=1" in net.last_launch()["kernel"]
    small = net.apply(P, x[:3000].contiguous())
    assert "GC=0" in net.last_launch()["kernel"]
    assert torch.equal(big[:3000], small)
    rows = np.arange(0, B, 97)
    xr = x.cpu().numpy()[rows]
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), xr.astype(np.float64))
    cancel, _ = _cancel(cfg, P, xr)
    assert (np.abs(big.cpu().numpy()[rows] - ref) <= 1e-5 * np.abs(ref) + 3e-6 * cancel).all()
    s0 = torch.from_numpy(configs.initial_state_from_query(x.cpu().numpy())).cuda()
    ctrl, states = planner.plan_tick(net, P, x, None, s0, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    assert "GC=1" in net.last_launch()["kernel"] and "ROLL=1" in net.last_launch()["kernel"]
    c2, s2 = planner.plan_tick(net, P, x[:3000].contiguous(), None, s0[:3000].contiguous(), configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    assert torch.equal(ctrl[:3000], c2) and torch.equal(states[:3000], s2) and torch.equal(ctrl, big)


@pytest.mark.parametrize("basis", ["gaussian", "inverse_multiquadric"])
def test_sparse_kernels_on_a_synthetic_grid_of_300_regions(gpu, basis):
    """Shapes the reference's fixtures do not reach: 300 regions (more than 256: 16-bit region lists; not a multiple of 32 or
    64), d = 3, O = 5, an odd number of centres per region (tail of the two-centre step), three split dimensions,
    dimension_ranges shorter than R (App. B-2: the last 12 regions stay 0).  Forward, tick-free VJP against the dense
    kernels and the float64 oracle; ragged batches."""
    import torch
    from oracle import c_oracle as co
    rng = np.random.default_rng(11)
    nx, ny, nz = 12, 12, 2                                    # 288 ranges for 300 regions, 26 gate factors (<= 32)
    ex, ey, ez = np.linspace(0.0, 9.0, nx + 1), np.linspace(-4.0, 4.0, ny + 1), np.linspace(-1.3, 1.3, nz + 1)
    cfg = {"in_features": 3, "out_features": 5, "num_kernels": 7, "basis_func": basis, "num_regions": 300,
           "activation_idx": [0, 1, 2], "delta": [40.0, 40.0, 15.0],
           "lower_bounds": [list(ex[:-1]), list(ey[:-1]), list(ez[:-1])], "upper_bounds": [list(ex[1:]), list(ey[1:]), list(ez[1:])],
           "dimension_ranges": [[i, j, k] for i in range(nx) for j in range(ny) for k in range(nz)]}
    R, K, D, O = 300, 7, 3, 5
    ctr = np.zeros((R, K, D))
    for r, (i, j, k) in enumerate(cfg["dimension_ranges"]):
        ctr[r] = np.stack([rng.uniform(ex[i], ex[i + 1], K), rng.uniform(ey[j], ey[j + 1], K), rng.uniform(ez[k], ez[k + 1], K)], axis=1)
    ctr[288:] = rng.normal(size=(12, K, D))
    P = {"params": {"rbf_list": {"centers": ctr.astype(np.float32), "log_sigs": rng.uniform(-1.0, 0.5, size=(R, K)).astype(np.float32)},
                    "linear": {"kernel": rng.normal(size=(K, O)).astype(np.float32), "bias": rng.normal(size=(O,)).astype(np.float32)}}}
    for B in (70, 1000, 4097):
        x = np.stack([rng.uniform(0, 9, B), rng.uniform(-4, 4, B), rng.uniform(-1.3, 1.3, B)], axis=1)
        x[::7, 0] = ex[rng.integers(1, nx, len(x[::7]))] + rng.choice([0.0, 0.01, -0.01], len(x[::7]))      # on / next to borders
        x = x.astype(np.float32)
        ref = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64))
        _, h, gamma = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), return_aux=True)
        cancel = np.abs(h) @ np.abs(P["params"]["linear"]["kernel"].astype(np.float64)) + 1e-30
        net = WCRBFNet.from_config(cfg)
        net.set_options(fwd_kernel=_lib.FWD_K1R)
        sp = net.apply(P, x)
        assert net.last_launch()["kernel"].startswith("rbf_fwd_sparse<D=3,OP=5")
        net.set_options(fwd_kernel=_lib.FWD_K1)
        dn = net.apply(P, x)
        assert (np.abs(sp - dn) <= 2e-6 * cancel + 1e-6 * np.abs(ref)).all(), (basis, B, np.abs(sp - dn).max())
        assert (np.abs(sp - ref) <= 1e-5 * np.abs(ref) + 3e-6 * cancel + 1e-6).all(), (basis, B)
        g = rng.normal(size=(B, O)).astype(np.float32)
        refg = co.wcrbf_vjp(cfg, P, x, g, np.float64)["params"]
        net.set_options(vjp_kernel=_lib.VJP_K2R)
        a = net.vjp(P, x, g)["params"]
        b = net.vjp(P, x, g)["params"]
        net.set_options(vjp_kernel=_lib.VJP_K2)
        d = net.vjp(P, x, g)["params"]
        for grp, name in LEAVES:
            assert np.array_equal(a[grp][name], b[grp][name])
            scale = np.abs(refg[grp][name]).max() + 1e-30
            assert np.abs(a[grp][name] - refg[grp][name]).max() <= 5e-5 * scale, (basis, B, name)
            assert np.abs(a[grp][name] - d[grp][name]).max() <= 2e-5 * scale, (basis, B, name)
        assert not a["rbf_list"]["centers"][288:].any() and not a["rbf_list"]["log_sigs"][288:].any()      # regions without a range
