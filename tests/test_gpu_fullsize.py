"""Every BASELINE configuration at its FULL size on one MI355X, against the oracle (the OpenMP C restatement in
float64 -- oracle/irbfn_oracle_impl.h, itself checked against the NumPy restatement in tests/test_oracle_cpu.py --
so that whole batches, not samples, are compared where that takes seconds) and through size-independent properties:
determinism, linearity of the VJP in the cotangent, independence of a result from its position in the batch,
fused == two-launch equality of the planning tick."""
import numpy as np
import pytest

from irbfn_amd import _lib, configs
from irbfn_amd import dynamics as dyn
from irbfn_amd.model import WCRBFNet
from irbfn_amd.planner import plan_batch
from oracle import c_oracle as co
from oracle import irbfn_oracle as orc
from test_gpu_parity import assert_states_close

import os

pytestmark = pytest.mark.gpu
RTOL = 1e-5          # north_star tolerance
_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_fullsize.txt")


def record(line: str):
    """The measured error figures of the full-size comparisons, printed AND kept: gpurun_out/parity_fullsize.txt on the GPU box
    (copied to profiles/r<NN>_parity_fullsize.txt for the round's evidence)."""
    print(line)
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


def _terms_scale(cfg, params, x):
    """sum_k |gamma phi_k W_ko| + |bias_o|: the magnitude of what the reduction adds up (float64, C oracle)."""
    p = params["params"]
    pa = {"params": {"rbf_list": p["rbf_list"], "linear": {"kernel": np.abs(p["linear"]["kernel"]), "bias": np.abs(p["linear"]["bias"])}}}
    return co.wcrbf_forward(cfg, pa, x, np.float64)


def test_cfg2_forward_full_batch(gpu):
    """Config 2: 4096 centres, d = 7, B = 65536 -- EVERY row against float64."""
    torch = gpu
    cfg, P, x = configs.model_card(2), configs.synth_params(2), configs.synth_queries(2)
    net = WCRBFNet.from_config(cfg)
    out = net.apply(P, torch.from_numpy(x).cuda())
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram<D=7,BC=0")               # the product path
    got = out.cpu().numpy().astype(np.float64)
    ref = co.wcrbf_forward(cfg, P, x, np.float64)
    scale = _terms_scale(cfg, P, x)
    err = np.abs(got - ref)
    with np.errstate(divide="ignore", invalid="ignore"):
        q999 = np.quantile(err / np.abs(ref), 0.999)
    record(f"cfg-2 full batch: max |err| / max |ref| = {err.max() / np.abs(ref).max():.2e}, max |err| / sum|terms| = {(err / scale).max():.2e}, "
          f"plain relative error: 99.9 % of the outputs below {q999:.2e}")
    assert err.max() <= RTOL * np.abs(ref).max()
    assert (err <= RTOL * np.abs(ref) + 3e-6 * scale).all()
    assert torch.equal(out, net.apply(P, torch.from_numpy(x).cuda()))                       # deterministic


def test_cfg3_vjp_full_batch(gpu):
    """Config 3: forward + VJP at B = 65536 -- all four gradient leaves against the float64 C oracle, determinism,
    exact homogeneity in the cotangent (power-of-two scaling) and additivity."""
    torch = gpu
    cfg, P = configs.model_card(3), configs.synth_params(3)
    x, g = configs.synth_queries(3), configs.synth_cotangent(3)
    net = WCRBFNet.from_config(cfg)
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    a = net.vjp(P, xt, gt)["params"]
    ref = co.wcrbf_vjp(cfg, P, x, g, np.float64)["params"]
    for grp, name in LEAVES:
        ga, gr = a[grp][name].cpu().numpy().astype(np.float64), ref[grp][name]
        scale = np.abs(gr).max()
        e = np.abs(ga - gr).max() / scale
        record(f"cfg-3 d {name}: max |err| / max |ref| = {e:.2e}")
        assert e <= 2e-5, (name, e)
    a2 = net.vjp(P, xt, gt)["params"]
    b = net.vjp(P, xt, gt * 4.0)["params"]
    g2 = torch.from_numpy(configs.synth_cotangent(3, seed=77)).cuda()
    c, d = net.vjp(P, xt, g2)["params"], net.vjp(P, xt, gt + g2)["params"]
    for grp, name in LEAVES:
        assert torch.equal(a[grp][name], a2[grp][name])                                     # bitwise reproducible
        assert torch.equal(b[grp][name], a[grp][name] * 4.0), name                          # exact: the scales are powers of two
        s = (a[grp][name] + c[grp][name] - d[grp][name]).abs().max() / d[grp][name].abs().max()
        assert float(s) <= 2e-5, (name, float(s))                                           # additive up to rounding


def test_cfg4_share_forward_and_tick(gpu):
    """Config 4, per-GPU share: N = 4096, O = 100 (T = 50), B = 32768 -- wide forward on 4096 rows against float64,
    the planning tick (forward -> 50-step ST-kinematic roll-out) against the oracle roll-out of the same controls on
    1024 rows, and fused tick == forward + stand-alone roll-out, bit for bit, on the whole batch."""
    torch = gpu
    cfg, P = configs.model_card(4), configs.synth_params(4)
    B = 32768
    x = configs.synth_queries(4, B=B)
    st0 = configs.initial_state_from_query(x)
    net = WCRBFNet.from_config(cfg)
    xt, st = torch.from_numpy(x).cuda(), torch.from_numpy(st0).cuda()
    ctrl, states = plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    assert net.last_launch()["kernel"].startswith(("rbf_tick_f16gram_wide", "rbf_tick_f16mfma_wide"))                  # ONE launch: forward + roll-out
    u = net.apply(P, xt)
    assert net.last_launch()["kernel"].startswith(("rbf_fwd_f16gram_wide", "rbf_fwd_f16mfma_wide"))
    assert torch.equal(ctrl, u)
    two = dyn.integrate_st_ks_mult(torch.cat([st, u], dim=1), configs.DYN_PARAMS)
    assert torch.equal(states, two)                                                         # fused == two launches
    net.set_options(tick_fused=0)
    ctrl2, states2 = plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    assert net.last_launch()["kernel"].startswith(("rbf_fwd_f16gram_wide", "rbf_fwd_f16mfma_wide"))
    net.set_options(tick_fused=1)
    assert torch.equal(ctrl2, ctrl) and torch.equal(states2, states)
    sub = np.arange(0, B, 8)                                                                # 4096 rows
    ref = co.wcrbf_forward(cfg, P, x[sub], np.float64)
    scale = _terms_scale(cfg, P, x[sub])
    err = np.abs(u.cpu().numpy()[sub].astype(np.float64) - ref)
    record(f"cfg-4 forward: max |err| / max |ref| = {err.max() / np.abs(ref).max():.2e}, / sum|terms| = {(err / scale).max():.2e}")
    assert err.max() <= RTOL * np.abs(ref).max() and (err <= RTOL * np.abs(ref) + 3e-6 * scale).all()
    rows = sub[:1024]
    dp = np.array(configs.DYN_PARAMS)
    # the roll-out is checked on ITS OWN input (the float32 controls the forward produced), in float64 and float32
    xu = np.hstack([st0[rows], u.cpu().numpy()[rows]])
    ref_states = orc.integrate_st_ks_mult(xu.astype(np.float64), dp)
    ref_states32 = orc.integrate_st_ks_mult(xu.astype(np.float32), dp.astype(np.float32))
    assert_states_close(states.cpu().numpy()[rows], ref_states, ref_states32)


def test_cfg4_rollout_whole_batch(gpu):
    """Config 4 roll-out at the WHOLE batch (262144 trajectories, T = 50): 16384 rows against the float64 / float32
    C oracle, bit-equality with the same trajectories rolled out at other positions of a smaller batch (tile
    boundaries, row alignment), determinism."""
    torch = gpu
    B, T = 262144, 50
    rng = np.random.default_rng(4)
    x = configs.synth_queries(4, B=B)
    xu = np.hstack([configs.initial_state_from_query(x), rng.normal(0, 2.0, size=(B, 2 * T)).astype(np.float32)])
    xt = torch.from_numpy(xu).cuda()
    for name, fn, kin in (("st_ks", dyn.integrate_st_ks_mult, True), ("st_select", dyn.integrate_st_mult, False)):
        s = fn(xt, configs.DYN_PARAMS)
        assert tuple(s.shape) == (B, T, 7)
        assert torch.equal(s, fn(xt, configs.DYN_PARAMS))
        rows = np.arange(5, B, 16)[:16384]
        dp = np.array(configs.DYN_PARAMS)
        ref = co.integrate_st_mult(xu[rows].astype(np.float64), dp, T, np.float64, kinematic_only=kin)
        ref32 = co.integrate_st_mult(xu[rows], dp.astype(np.float32), T, np.float32, kinematic_only=kin)
        assert_states_close(s.cpu().numpy()[rows], ref, ref32)
        # the same trajectories at other batch positions / another batch size (ragged last tile): identical bits
        pick = torch.from_numpy(np.concatenate([np.arange(1000, 1000 + 777), np.arange(B - 100, B)])).cuda()
        s2 = fn(xt[pick].contiguous(), configs.DYN_PARAMS)
        assert torch.equal(s2, s[pick]), name


def test_cfg5_forward_full_batch(gpu):
    """Config 5: 16384 inverse-multiquadric centres, B = 2^20 -- 2048 rows against float64 and position independence
    (a query gives the same bits wherever it sits in whatever batch)."""
    torch = gpu
    cfg, P = configs.model_card(5), configs.synth_params(5)
    x = configs.synth_queries(5)
    assert x.shape[0] == 1 << 20
    net = WCRBFNet.from_config(cfg)
    xt = torch.from_numpy(x).cuda()
    out = net.apply(P, xt)
    assert net.last_launch()["kernel"].startswith("rbf_fwd_f16gram<D=7,BC=2")
    rows = np.arange(0, 1 << 20, 512)
    ref = co.wcrbf_forward(cfg, P, x[rows], np.float64)
    scale = _terms_scale(cfg, P, x[rows])
    err = np.abs(out.cpu().numpy()[rows].astype(np.float64) - ref)
    record(f"cfg-5: max |err| / max |ref| = {err.max() / np.abs(ref).max():.2e}, / sum|terms| = {(err / scale).max():.2e}")
    assert err.max() <= RTOL * np.abs(ref).max() and (err <= RTOL * np.abs(ref) + 3e-6 * scale).all()
    perm = torch.from_numpy(np.random.default_rng(0).permutation(1 << 20)).cuda()          # same batch size = same launch geometry
    assert torch.equal(net.apply(P, xt[perm].contiguous()), out[perm])


@pytest.mark.parametrize("B", [32768 + 77, 4099, 262144])
def test_cfg4_fused_tick_mirror_and_ragged(gpu, B):
    """The one-launch planning tick (plan_tick_wide.hip) with the planner's mirror flags, on batches that end inside a
    block / a wave (ragged) and on the whole config-4 batch, for both single-track modes: controls and states equal,
    bit for bit, to forward -> sign flip -> stand-alone roll-out; states without a controls buffer."""
    torch = gpu
    from irbfn_amd.planner import plan_tick
    cfg, P = configs.model_card(4), configs.synth_params(4)
    x = configs.synth_queries(4, B=B)
    st0 = configs.initial_state_from_query(x)
    mirror = (np.random.default_rng(B).random(B) < 0.5).astype(np.int32)
    net = WCRBFNet.from_config(cfg)
    xt, st, mt = torch.from_numpy(x).cuda(), torch.from_numpy(st0).cuda(), torch.from_numpy(mirror).cuda()
    T = cfg["out_features"] // 2
    for mode, fn in ((_lib.ROLLOUT_ST_KS, dyn.integrate_st_ks_mult), (_lib.ROLLOUT_ST_SELECT, dyn.integrate_st_mult)):
        ctrl, states = plan_tick(net, P, xt, mt, st, configs.DYN_PARAMS, mode=mode)
        # small batches run 4 centre slices per block (the pipelined kernel's LDS ring does not fit): separate launches
        assert net.last_launch()["kernel"].startswith(("rbf_tick_f16gram_wide", "rbf_tick_f16mfma_wide") if B > 8192 else "rollout_fwd") or B <= 8192
        u = net.apply(P, xt).clone()
        u[:, T:] = torch.where(mt[:, None] != 0, -u[:, T:], u[:, T:])
        assert torch.equal(ctrl, u)
        assert torch.equal(states, fn(torch.cat([st, u], dim=1), configs.DYN_PARAMS))
        _, only_states = plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=mode, return_controls=False)
        ref = fn(torch.cat([st, net.apply(P, xt)], dim=1), configs.DYN_PARAMS)
        assert torch.equal(only_states, ref)


@pytest.mark.parametrize("basis,T", [("inverse_quadratic", 50), ("inverse_multiquadric", 50), ("gaussian", 49), ("gaussian_wide", 53)])
def test_cfg4_fused_tick_other_instances(gpu, basis, T):
    """The other instances of the one-launch tick (inverse-quadratic / inverse-multiquadric basis, T = 49: O = 98 in the
    same seven column tiles) against forward -> stand-alone roll-out, bit for bit; T = 53 (O = 106) has no instance
    (the lanes hold 50 control knots) and must take the separate launches with the same results."""
    torch = gpu
    cfg = dict(configs.model_card(4), basis_func=basis, out_features=2 * T)
    P = configs.synth_params(4)
    rng = np.random.default_rng(T)
    P["params"]["linear"] = {"kernel": rng.normal(0.0, 0.05, size=(cfg["num_kernels"], 2 * T)).astype(np.float32),
                             "bias": rng.normal(0.0, 0.1, size=(2 * T,)).astype(np.float32)}
    B = 33000
    x = configs.synth_queries(4, B=B)
    st0 = configs.initial_state_from_query(x)
    net = WCRBFNet.from_config(cfg)
    xt, st = torch.from_numpy(x).cuda(), torch.from_numpy(st0).cuda()
    ctrl, states = plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    fused = net.last_launch()["kernel"].startswith(("rbf_tick_f16gram_wide", "rbf_tick_f16mfma_wide"))
    assert fused == (T <= 50)
    u = net.apply(P, xt)
    assert torch.equal(ctrl, u)
    assert tuple(states.shape) == (B, T, 7)
    assert torch.equal(states, dyn.integrate_st_ks_mult(torch.cat([st, u], dim=1), configs.DYN_PARAMS))
    ref = co.wcrbf_forward(cfg, P, x[:512], np.float64)
    err = np.abs(u.cpu().numpy()[:512].astype(np.float64) - ref)
    assert err.max() <= RTOL * np.abs(ref).max()
