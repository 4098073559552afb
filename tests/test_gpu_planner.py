"""GPU parity of the batched planner front / back end against the literal per-row
restatements in oracle/irbfn_oracle.py (irbfn_planner.py:181-212, :456-502; explicit_planner.py:165-175,
:383).  Index / flag outputs must be bit-exact; float32 query columns equal NumPy's float64-then-cast
values up to 1 ulp of float32 (device libm vs NumPy's in the rotation)."""
import numpy as np
import pytest
import torch

from conftest import load_ckpt_fixture
from irbfn_amd import _lib, configs, explicit_planner, planner
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu
DP = np.array(configs.DYN_PARAMS)


def _poses(rng, B):
    pose = np.stack([rng.uniform(-50, 50, B), rng.uniform(-50, 50, B), rng.uniform(-0.4, 0.4, B), rng.uniform(0, 7, B),
                     rng.uniform(-3.2, 3.2, B), rng.uniform(-2, 2, B), rng.uniform(-0.3, 0.3, B)], axis=1)
    d = rng.uniform(0.3, 3.5, B)
    ang = pose[:, 4] + rng.uniform(-1.2, 1.2, B)
    goal = np.stack([pose[:, 0] + d * np.cos(ang), pose[:, 1] + d * np.sin(ang), pose[:, 4] + rng.uniform(-4, 4, B),
                     rng.uniform(0.5, 7, B)], axis=1)
    return pose, goal


@pytest.mark.parametrize("B", [1, 257])
def test_cartesian_queries_and_mirror_flags(gpu, B):
    rng = np.random.default_rng(B)
    pose, goal = _poses(rng, B)
    if B > 4:
        goal[3, :2] = pose[3, :2] + [1.0, 0.0]                   # goal straight along +x of the WORLD frame
        pose[3, 4] = 0.0                                          # ... and heading 0: goal_local[1] == 0 -> no mirror
        goal[4, 2] = pose[4, 4]                                   # goal_theta == 0 -> 0 % pi == 0
    x, s0, mirror = planner.build_queries(pose, goal)
    ref = [orc.plan_query_cartesian(pose[b], goal[b]) for b in range(B)]
    rx = np.stack([r[0] for r in ref]); rm = np.array([r[1] for r in ref]); rs = np.stack([r[2] for r in ref])
    np.testing.assert_array_equal(mirror.cpu().numpy().astype(bool), rm)
    np.testing.assert_array_equal(s0.cpu().numpy(), rs)
    gx = x.cpu().numpy()
    np.testing.assert_array_equal(gx[:, [0, 4, 5, 6]], rx[:, [0, 4, 5, 6]])
    # rotated columns: <= 1 float32 ulp at the magnitude of the goal offset
    assert np.abs(gx[:, 1:4] - rx[:, 1:4]).max() <= 5e-7
    assert (gx[:, 2] >= 0).all() and (gx[:, 3] >= 0).all() and (gx[:, 3] < np.float32(np.pi) + 1e-6).all()
    assert 0 < rm.sum() < B or B == 1


def test_frenet_queries(gpu):
    rng = np.random.default_rng(3)
    B = 300
    fr = np.stack([rng.uniform(0, 100, B), rng.uniform(-0.8, 0.8, B), rng.uniform(-0.4, 0.4, B), rng.uniform(0.5, 7, B),
                   rng.uniform(-1, 1, B), rng.uniform(-2, 2, B), rng.uniform(-0.6, 0.6, B), rng.uniform(-0.5, 0.5, B)], axis=1)
    fr[0, 1] = -0.05          # boundary: ey < -0.05 is strict
    fr[1, 1] = -0.0500001
    vg = rng.uniform(0.5, 7, B)
    x, s0, mirror = planner.build_queries_frenet(fr, vg)
    ref = [orc.plan_query_frenet(fr[b], vg[b]) for b in range(B)]
    np.testing.assert_array_equal(x.cpu().numpy(), np.stack([r[0] for r in ref]))
    np.testing.assert_array_equal(mirror.cpu().numpy().astype(bool), np.array([r[1] for r in ref]))
    np.testing.assert_array_equal(s0.cpu().numpy(), np.stack([r[2] for r in ref]))
    assert not mirror[0] and mirror[1]


@pytest.mark.parametrize("B", [1, 40, 500])
def test_plan_tick_unmirrors_and_rolls_out(gpu, B):
    """Full batched IRBFNPlanner.plan data path on the reference's trained 5-step checkpoint: queries ->
    pred_step -> sign flip of sv controls of mirrored rows -> integrate_st_mult from the true poses."""
    cfg, params, *_ = load_ckpt_fixture("dnmpc_1regions_newdata_oldintloss_nomirror_highk")
    net = WCRBFNet.from_config(cfg)
    rng = np.random.default_rng(10 + B)
    pose, goal = _poses(rng, B)
    x, s0, mirror = planner.build_queries(pose, goal)
    ctrl, states = planner.plan_tick(net, params, x, mirror, s0, DP, mode=_lib.ROLLOUT_ST_SELECT)
    x_np, m_np = x.cpu().numpy(), mirror.cpu().numpy()
    p64 = orc.cast_params(params, np.float64)
    pred = orc.wcrbfnet_apply(cfg, p64, x_np.astype(np.float64))
    ref_u = orc.unmirror_controls(pred, m_np, 5)
    got_u = ctrl.cpu().numpy()
    scale = np.abs(ref_u).max()
    assert np.abs(got_u - ref_u).max() <= 2e-5 * scale + 1e-5
    # the flip itself is exact: a second tick without flags differs only by the sign of [5:] on mirrored rows
    ctrl0, _ = planner.plan_tick(net, params, x, None, s0, DP, mode=_lib.ROLLOUT_ST_SELECT)
    np.testing.assert_array_equal(got_u, orc.unmirror_controls(ctrl0.cpu().numpy(), m_np, 5))
    # roll-out from float32(pose) with the device's own controls (isolates the integrator)
    ref_states = orc.integrate_st_mult(np.hstack([s0.cpu().numpy().astype(np.float64), got_u.astype(np.float64)]), DP)
    err = np.abs(states.cpu().numpy() - ref_states)
    assert (err <= 2e-5 * (1 + np.abs(ref_states))).all(), err.max()
    # controls only (no roll-out)
    ctrl2, none = planner.plan_tick(net, params, x, mirror, rollout=False)
    assert none is None
    assert np.abs(ctrl2.cpu().numpy() - ref_u).max() <= 2e-5 * scale + 1e-5


def _grid(rng, axes, T=5):
    inputs = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, len(axes))
    outputs = rng.normal(size=(inputs.shape[0], T, 2))
    return inputs, outputs


def test_lut_grid_lookup_bit_exact(gpu):
    rng = np.random.default_rng(0)
    axes = [np.linspace(0.5, 7.0, 8), np.linspace(0.0, 3.6, 9), np.linspace(0.0, 3.6, 7), np.linspace(0.0, 3.1, 6),
            np.linspace(0.5, 7.0, 4), np.linspace(-0.4, 0.4, 3), np.linspace(-2.0, 2.0, 5)]
    inputs, outputs = _grid(rng, axes)
    tab = explicit_planner.ExplicitTable(inputs, outputs)
    assert tab.is_grid and tab.shape == [8, 9, 7, 6, 4, 3, 5]
    B = 2000
    lo = np.array([a[0] for a in axes]) - 0.5
    hi = np.array([a[-1] for a in axes]) + 0.5
    q = rng.uniform(lo, hi, size=(B, 7))
    q[:50] = inputs[rng.integers(0, inputs.shape[0], 50)]        # exactly on grid values: side="right" matters
    q[50, 0] = np.nan
    idx, out = tab.grid_lookup(q)
    ref_idx = np.array([orc.lut_grid_lookup(tab.input_keys, tab.shape, q[b])[1] for b in range(B)])
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    ref_out = outputs.reshape((*tab.shape, -1))[tuple(np.unravel_index(ref_idx, tab.shape))]
    np.testing.assert_array_equal(out.cpu().numpy(), ref_out.astype(np.float32))
    ragged = explicit_planner.ExplicitTable(inputs[:-1], outputs[:-1])
    with pytest.raises(ValueError):
        ragged.grid_lookup(q)


@pytest.mark.parametrize("D,N,B", [(8, 50000, 37), (7, 1000, 1), (3, 300000, 9)])
def test_lut_nearest_matches_kdtree(gpu, D, N, B):
    """explicit_planner.py:383 -- scipy's KDTree is the reference's own look-up and importable here."""
    rng = np.random.default_rng(N)
    inputs = rng.uniform(-3, 3, size=(N, D)).astype(np.float32)
    outputs = rng.normal(size=(N, 5, 2))
    q = rng.uniform(-3.2, 3.2, size=(B, D)).astype(np.float32)
    q[0] = inputs[N // 2]                                         # exact hit -> distance 0
    tab = explicit_planner.ExplicitTable(inputs, outputs)
    idx, dist, out = tab.nearest(q)
    rd, ri = orc.lut_nearest(inputs.astype(np.float64), q.astype(np.float64))
    gi = idx.cpu().numpy()
    same = gi == ri
    # a different row is acceptable only as a float32 near-tie: its true distance equals the minimum to 1e-6
    true_d = np.linalg.norm(inputs[gi].astype(np.float64) - q.astype(np.float64), axis=1)
    assert (true_d <= rd * (1 + 1e-6) + 1e-7).all()
    assert same.mean() >= 0.97 and gi[0] == N // 2 and float(dist[0]) == 0.0
    np.testing.assert_allclose(dist.cpu().numpy(), rd, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out.cpu().numpy(), outputs.reshape(N, -1).astype(np.float32)[gi])
    # duplicate rows: ties resolve to the lowest index
    dup = np.concatenate([inputs[:100], inputs[:100]])
    t2 = explicit_planner.ExplicitTable(dup, np.zeros((200, 5, 2)))
    i2, *_ = t2.nearest(inputs[:100])
    np.testing.assert_array_equal(i2.cpu().numpy(), np.arange(100))


def test_waypoint_geometry_matches_the_restatement(gpu):
    """nearest_point / intersect_point (planner_utils.py:109-233), batched, against the NumPy statement of the same
    lines point by point (numba semantics themselves: parity unpinned)."""
    from irbfn_amd import planner_utils as pu
    rng = np.random.default_rng(8)
    th = np.linspace(0, 2 * np.pi, 200, endpoint=False)
    traj = np.stack([20 * np.cos(th) + 3 * np.cos(3 * th), 12 * np.sin(th)], axis=1)        # a closed race line
    pts = traj[rng.integers(0, 200, size=500)] + rng.normal(size=(500, 2)) * 0.8
    proj, dist, t, seg = (v.cpu().numpy() for v in pu.nearest_point(pts, traj))
    for b in range(500):
        rp, rd, rt, ri = orc.nearest_point(pts[b], traj)
        assert seg[b] == ri and abs(dist[b] - rd) <= 1e-6 and abs(t[b] - rt) <= 1e-5 and np.abs(proj[b] - rp).max() <= 1e-5
    start = seg + t
    for wrap, radius in ((False, 1.5), (True, 1.5), (True, 60.0)):
        fp, fi, ft, found = (v.cpu().numpy() for v in pu.intersect_point(pts, radius, traj, start, wrap=wrap))
        nfound = 0
        for b in range(500):
            rp, ri, rt = orc.intersect_point(pts[b], radius, traj, float(start[b]), wrap)
            if rp is None:
                assert found[b] == 0 and np.isnan(fp[b]).all()
            else:
                nfound += 1
                assert found[b] == 1 and fi[b] == ri and abs(ft[b] - rt) <= 2e-4 and np.abs(fp[b] - rp).max() <= 1e-3
        assert (nfound > 400) if radius < 10 else (nfound == 0)      # a 60 m look-ahead circle misses a 23 m track


@pytest.mark.gpu
@pytest.mark.parametrize("mode_name,D,T,B,basis", [
    ("st_ks", 7, 5, 4099, "gaussian"), ("st_select", 7, 5, 70000, "gaussian"), ("st_ks", 7, 1, 1000, "gaussian"),
    ("st_ks", 7, 8, 333, "inverse_quadratic"), ("fullint", 7, 5, 2500, "inverse_multiquadric"), ("frenet", 8, 5, 4099, "gaussian"),
    ("frenet", 8, 5, 129, "inverse_quadratic"), ("frenet", 8, 2, 20000, "inverse_multiquadric"),
    ("fullint", 7, 5, 80000, "gaussian"), ("frenet", 8, 5, 66000, "gaussian")])       # 2048 < B / 32 <= 3072: blocks of four waves (S = 1, QG = 4)
def test_narrow_tick_in_one_launch(gpu, mode_name, D, T, B, basis):
    """The planning tick of the narrow nets on the matrix-core kernel (`rbf_tick_f16mfma`: forward + sign flip + roll-out
    by the wave that produced the rows): controls and states equal, bit for bit, the forward followed by the stand-alone
    roll-out kernel -- all four models, T = 1 / 5 / 8, ragged batches, mirror flags, with and without a controls buffer;
    the forward itself against the float64 restatement."""
    import torch
    from irbfn_amd import _lib, configs, dynamics as dyn
    from irbfn_amd.model import WCRBFNet
    from irbfn_amd.planner import plan_batch, plan_tick
    from oracle import c_oracle as co
    rng = np.random.default_rng(B + T)
    K, O = 512, 2 * T
    lo, hi = [-1.0] * D, [1.0] * D
    cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": basis, "num_regions": 1,
           "lower_bounds": [[v] for v in lo], "upper_bounds": [[v] for v in hi], "dimension_ranges": [[0] * D],
           "activation_idx": list(range(D)), "delta": [5.0] * D}
    P = {"params": {"rbf_list": {"centers": rng.uniform(-1.2, 1.2, size=(1, K, D)).astype(np.float32),
                                 "log_sigs": rng.uniform(-0.5, 0.3, size=(1, K)).astype(np.float32)},
                    "linear": {"kernel": (rng.normal(size=(K, O)) * 0.2).astype(np.float32),
                               "bias": (rng.normal(size=(O,)) * 0.1).astype(np.float32)}}}
    x = rng.uniform(-1, 1, size=(B, D)).astype(np.float32)
    mode = {"st_ks": _lib.ROLLOUT_ST_KS, "st_select": _lib.ROLLOUT_ST_SELECT, "fullint": _lib.ROLLOUT_FULLINT,
            "frenet": _lib.ROLLOUT_FRENET_LS}[mode_name]
    if mode_name == "fullint":
        st0 = rng.uniform(0.5, 6.0, size=(B, 1)).astype(np.float32)
    elif mode_name == "frenet":
        st0 = np.hstack([rng.normal(size=(B, 1)) * 0.2, rng.normal(size=(B, 1)) * 0.2, rng.normal(size=(B, 1)) * 0.1,
                         rng.uniform(1, 6, size=(B, 1)), rng.normal(size=(B, 3)) * 0.1, rng.normal(size=(B, 1)) * 0.05]).astype(np.float32)
    else:
        st0 = np.hstack([rng.normal(size=(B, 3)) * 0.3, rng.uniform(0.5, 7.0, size=(B, 1)), rng.normal(size=(B, 3)) * 0.2]).astype(np.float32)
    mirror = (rng.random(B) < 0.5).astype(np.int32)
    net = WCRBFNet.from_config(cfg)
    xt, st, mt = torch.from_numpy(x).cuda(), torch.from_numpy(st0).cuda(), torch.from_numpy(mirror).cuda()
    ctrl, states = plan_tick(net, P, xt, mt, st, configs.DYN_PARAMS, mode=mode)
    tick_kernel = net.last_launch()["kernel"]               # K1g where the parameters fit its expansion and d = 7, else K1h
    assert tick_kernel.startswith("rbf_tick_f16gram<") or tick_kernel.startswith("rbf_tick_f16mfma<")
    if tick_kernel.startswith("rbf_tick_f16gram<") and 2048 * 32 < B <= 3072 * 32:
        assert "S=1,QG=4" in tick_kernel, tick_kernel
    u = net.apply(P, xt).clone()
    assert net.last_launch()["kernel"].startswith(tick_kernel.split("<")[0].replace("tick", "fwd") + "<")
    ref = co.wcrbf_forward(cfg, P, x, np.float64)
    assert np.abs(u.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    u[:, T:] = torch.where(mt[:, None] != 0, -u[:, T:], u[:, T:])
    assert torch.equal(ctrl, u)
    x0u = torch.cat([st, u], dim=1)
    two = dyn.rollout_forward(mode, x0u, configs.DYN_PARAMS if mode_name != "fullint" else None, T)
    assert tuple(states.shape) == tuple(two.shape) and torch.equal(states, two)
    _, s2 = plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=mode, return_controls=False)          # no controls buffer
    assert net.last_launch()["kernel"] == tick_kernel
    plain = dyn.rollout_forward(mode, torch.cat([st, net.apply(P, xt)], dim=1), configs.DYN_PARAMS if mode_name != "fullint" else None, T)
    assert torch.equal(s2, plain)
    net.set_options(tick_fused=0)                                                                    # forward -> roll-out launches
    c3, s3 = plan_tick(net, P, xt, mt, st, configs.DYN_PARAMS, mode=mode)
    assert not net.last_launch()["kernel"].startswith("rbf_tick")
    assert torch.equal(c3, ctrl) and torch.equal(s3, states)
