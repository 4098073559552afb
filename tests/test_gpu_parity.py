"""GPU parity tests (run on the MI355X box with ``-m gpu``): the HIP path, called through the
C-ABI library, against the CPU oracle on the same seeded inputs, against the committed golden
fixtures, and -- at BASELINE sizes -- through size-independent properties.

Tolerances.  north_star asks for 1e-5 relative on predicted controls / trajectory states.  The
float64 oracle is the reference value; the product computes in float32 like the reference's default
(JAX float32), so for *trained* nets whose Dense kernel reaches +-800 with O(1) outputs the error is
bounded relative to the magnitude of the summed terms sum_k |h_k W_ko|, and it is
also required not to exceed the error of a plain float32 CPU evaluation by more than a small factor.
"""
import numpy as np
import pytest

from conftest import CKPT_RUNS, load_ckpt_fixture, load_deeper_fixture
from irbfn_amd import _lib, configs
from irbfn_amd import dynamics as dyn
from irbfn_amd import planner_utils as pu
from irbfn_amd.model import WCRBFNet, make_state, pred_step
from oracle import c_oracle as co
from oracle import hand_vjp as hv
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu
RTOL = 1e-5          # north_star tolerance
DP = np.array(configs.DYN_PARAMS)


def relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def assert_states_close(got, ref, ref32=None, rtol=RTOL):
    """Trajectory states against the float64 oracle.  Pass if |err| <= rtol * (|ref| + scale), scale =
    max |ref| of that state component along the trajectory -- or, where float32 integration itself is
    worse conditioned than that (T = 50 steps; the dynamic single-track RHS has a prefactor
    mu*m/(I*L) = 67 and 1/V terms), if the error is within 4x the error the float32 NumPy restatement
    of the reference makes on the same trajectory component (the reference itself runs in float32)."""
    got = np.asarray(got, np.float64)
    assert got.shape == ref.shape
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-3)
    err = np.abs(got - ref)
    bound = rtol * (np.abs(ref) + scale)
    if ref32 is not None:
        e32 = np.abs(np.asarray(ref32, np.float64) - ref).max(axis=1, keepdims=True)
        bound = np.maximum(bound, 4.0 * e32 + 1e-7 * scale)
    bad = err > bound
    assert not bad.any(), (int(bad.sum()), float(err.max()), float((err / (np.abs(ref) + scale)).max()))


F32 = np.float32
DP32 = np.array(configs.DYN_PARAMS, np.float32)


def _p32(params):
    return orc.cast_params(params, np.float32)


# ------------------------------------------------------------------ forward
def test_native_library_is_loaded(gpu):
    lib = _lib.load()
    assert lib.irbfn_device_count() >= 1
    import os
    maps = open("/proc/self/maps").read()
    assert os.path.join("irbfn_amd", "libirbfn_hip.so") in maps


@pytest.mark.parametrize("run", CKPT_RUNS)
def test_forward_trained_checkpoints(gpu, run):
    cfg, params, x, out64, h64, gamma64 = load_ckpt_fixture(run)
    net = WCRBFNet.from_config(cfg)
    out = net.apply(params, x.astype(np.float32))
    assert out.shape == out64.shape and out.dtype == np.float32
    W = np.abs(np.asarray(params["params"]["linear"]["kernel"], np.float64))
    scale = np.abs(h64) @ W                                  # magnitude of the summed terms
    err = np.abs(out.astype(np.float64) - out64)
    assert (err <= RTOL * np.abs(out64) + 3e-6 * scale).all(), (err.max(), scale.max())
    ref32 = co.wcrbf_forward(cfg, params, x, np.float32)     # a plain float32 CPU evaluation
    e32 = np.abs(ref32.astype(np.float64) - out64)
    assert err.max() <= 4 * e32.max() + 1e-6 * scale.max()
    gam = net.gate(x.astype(np.float32))
    np.testing.assert_allclose(gam, gamma64, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("run", CKPT_RUNS)
def test_small_batch_kernel_and_tiled_kernel_agree(gpu, run):
    """B <= 64 dispatches to the centre-lane latency kernel (K1s); the option fwd_kernel = K1 forces the tiled
    query-lane kernel (K1, incl. its gated R > 1 path).  Both must meet the oracle and each other."""
    cfg, params, x, out64, h64, _ = load_ckpt_fixture(run)
    net = WCRBFNet.from_config(cfg)
    x32 = x.astype(np.float32)
    a = net.apply(params, x32)
    assert "clane" in net.last_launch()["kernel"]
    one = net.apply(params, x32[:1])                          # the planner's B = 1 call
    np.testing.assert_array_equal(one, net.apply(params, x32[:1]))   # fixed-order reduction: deterministic
    net.set_options(fwd_kernel=_lib.FWD_K1)
    b = net.apply(params, x32)
    assert "qlane" in net.last_launch()["kernel"]
    scale = np.abs(h64) @ np.abs(np.asarray(params["params"]["linear"]["kernel"], np.float64))
    for out in (a, b):
        assert (np.abs(out - out64) <= RTOL * np.abs(out64) + 3e-6 * scale).all()
    assert (np.abs(a - b) <= 4e-6 * scale + 1e-6).all()
    assert (np.abs(one - a[:1]) <= 4e-6 * scale[:1] + 1e-6).all()


def test_forward_cfg1_fixture(gpu):
    import os
    from conftest import GOLDEN
    cfg = configs.model_card(1)
    net = WCRBFNet.from_config(cfg)
    out = net.apply(configs.synth_params(1), configs.synth_queries(1))
    exp = np.load(os.path.join(GOLDEN, "synth_cfg1.npz"))["out64"]
    assert relmax(out, exp) <= RTOL


@pytest.mark.parametrize("basis", sorted(orc.BASIS))
def test_forward_all_bases(gpu, basis):
    cfg = dict(configs.model_card(1), basis_func=basis)
    P = configs.synth_params(1)
    x = configs.synth_queries(1, B=257)
    exp = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64))
    out = WCRBFNet.from_config(cfg).apply(P, x)
    assert relmax(out, exp) <= 2 * RTOL, basis


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_forward_ragged_batches(gpu, B):
    cfg = configs.model_card(2)
    P = configs.synth_params(2)
    net = WCRBFNet.from_config(cfg)
    x = configs.synth_queries(2, B=B)
    exp = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64), chunk=256)
    assert relmax(net.apply(P, x), exp) <= RTOL


def test_forward_empty_batch_and_errors(gpu):
    cfg = configs.model_card(1)
    net = WCRBFNet.from_config(cfg)
    P = configs.synth_params(1)
    assert net.apply(P, np.zeros((0, 3), np.float32)).shape == (0, 5)
    with pytest.raises(ValueError):
        net.apply(P, np.zeros((4, 4), np.float32))
    with pytest.raises(ValueError):
        WCRBFNet.from_config(cfg)(np.zeros((4, 3), np.float32))          # nothing bound
    with pytest.raises(ValueError):
        net.apply(configs.synth_params(2), np.zeros((4, 3), np.float32))  # wrong param shapes
    with pytest.raises(ValueError):
        WCRBFNet.from_config(dict(cfg, in_features=9, activation_idx=[0], lower_bounds=[[0.0]],
                                  upper_bounds=[[1.0]], delta=[1.0], dimension_ranges=[[0]])
                             ).gate(np.zeros((2, 9), np.float32))        # D > 8: outside the kernel set


@pytest.mark.parametrize("D,O,R,K", [(2, 3, 1, 50), (5, 7, 1, 33), (6, 20, 2, 17), (8, 1, 3, 40), (4, 12, 1, 64),
                                     (7, 100, 1, 96), (3, 128, 1, 70)])
def test_forward_padded_shapes(gpu, D, O, R, K):
    rng = np.random.default_rng(D * 100 + O)
    ns = min(D, 3)
    cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": "inverse_quadratic",
           "num_regions": R, "activation_idx": list(range(ns)), "delta": [5.0] * ns,
           "lower_bounds": [[-2.0, 0.0][:max(1, min(2, R))] for _ in range(ns)],
           "upper_bounds": [[0.0, 2.0][:max(1, min(2, R))] for _ in range(ns)],
           "dimension_ranges": [[(r >> d) & 1 if R > 1 else 0 for d in range(ns)] for r in range(R)]}
    P = {"params": {"rbf_list": {"centers": rng.normal(size=(R, K, D)), "log_sigs": rng.uniform(-0.5, 1, size=(R, K))},
                    "linear": {"kernel": rng.normal(size=(K, O)), "bias": rng.normal(size=(O,))}}}
    x = rng.uniform(-2, 2, size=(200, D))
    exp = orc.wcrbfnet_apply(cfg, P, x)
    out = WCRBFNet.from_config(cfg).apply(P, x.astype(np.float32))
    assert relmax(out, exp) <= 2 * RTOL


def test_forward_regions_without_range_and_far_regions(gpu):
    cfg, params, x, out64, *_ = load_ckpt_fixture("dnmpc_128regions")
    cfg = dict(cfg, dimension_ranges=cfg["dimension_ranges"][:100])      # App. B-2: 28 regions stay 0
    exp = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x)
    out = WCRBFNet.from_config(cfg).apply(params, x.astype(np.float32))
    # north_star's 1e-5, relative to the output scale, + the cancellation term of the trained Dense layer (weights up to
    # +-480 against outputs O(1)) exactly as for the full 128-range card
    _, h, _ = orc.wcrbfnet_apply(cfg, orc.cast_params(params, np.float64), x, return_aux=True)
    cancel = np.abs(h) @ np.abs(np.asarray(params["params"]["linear"]["kernel"], np.float64))
    assert (np.abs(out - exp) <= 1e-5 * np.abs(exp).max() + 3e-6 * cancel).all()
    assert np.abs(out - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max())


def test_forward_nan_and_inf_propagate(gpu):
    cfg = configs.model_card(1)
    net = WCRBFNet.from_config(cfg)
    x = configs.synth_queries(1, B=70)
    x[3, 1] = np.nan
    x[5, 0] = np.inf
    P = configs.synth_params(1)
    out = net.apply(P, x)
    assert np.isnan(out[3]).all() and np.isfinite(out[[0, 1, 2, 4, 6]]).all()
    # an infinite coordinate: r2 = inf -> phi = exp(-inf) = 0 for every centre, the gate's tanh saturates (0 or 1), so
    # the row is IEEE-finite and equals the bias times the gate -- exactly what the float64 restatement returns
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64))
    assert np.isnan(ref[3]).all() and np.isfinite(ref[5]).all()
    assert np.isfinite(out[5]).all() and np.abs(out[5] - ref[5]).max() <= 1e-5 * max(1.0, np.abs(ref[5]).max())
    x[9, 2] = -np.inf                                         # the other side of the gate: gamma = 0 -> bias * 0
    out2 = net.apply(P, x)
    ref2 = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64))
    assert np.isfinite(out2[9]).all() == np.isfinite(ref2[9]).all()
    assert np.allclose(out2[9], ref2[9], rtol=1e-5, atol=1e-6, equal_nan=True)


def test_torch_tensor_inputs_and_pred_step(gpu):
    torch = gpu
    cfg = configs.model_card(2)
    net = WCRBFNet.from_config(cfg)
    P = configs.synth_params(2)
    x = configs.synth_queries(2, B=300)
    xt = torch.from_numpy(x).cuda()
    Pt = {"params": {"rbf_list": {k: torch.from_numpy(v).cuda() for k, v in P["params"]["rbf_list"].items()},
                     "linear": {k: torch.from_numpy(v).cuda() for k, v in P["params"]["linear"].items()}}}
    out = pred_step(make_state(net, Pt), xt)
    assert out.is_cuda and out.dtype == torch.float32
    np.testing.assert_array_equal(out.cpu().numpy(), net.apply(P, x))
    # in-place parameter update is picked up (optimizer step): version counter changes the fingerprint
    Pt["params"]["linear"]["bias"].add_(1.0)
    out2 = net.apply(Pt, xt)
    np.testing.assert_allclose(out2.cpu().numpy(), out.cpu().numpy() + 1.0, rtol=0, atol=1e-5)


# ------------------------------------------------------------------ BASELINE sizes: properties
def test_cfg2_full_size_properties(gpu):
    torch = gpu
    cfg = configs.model_card(2)
    P = configs.synth_params(2)
    net = WCRBFNet.from_config(cfg)
    x = configs.synth_queries(2)                               # B = 65536
    xt = torch.from_numpy(x).cuda()
    out = net.apply(P, xt)
    assert out.shape == (65536, 10)
    # (1) parity on a 1024-query subset
    idx = np.random.default_rng(7).choice(65536, 1024, replace=False)
    exp = co.wcrbf_forward(cfg, P, x[idx], np.float64)
    assert relmax(out.cpu().numpy()[idx], exp) <= RTOL
    # (2) determinism (fixed-order reductions): bitwise identical reruns
    assert torch.equal(out, net.apply(P, xt))
    # (3) a query's result does not depend on its position in the batch
    perm = torch.from_numpy(np.random.default_rng(8).permutation(65536)).cuda()
    assert torch.equal(net.apply(P, xt[perm].contiguous()), out[perm])
    # (4) linearity in the Dense layer: net(W1 + W2, b1 + b2) = net(W1, b1) + net(W2, b2)
    P2 = configs.synth_params(2)
    rng = np.random.default_rng(9)
    P2["params"]["linear"]["kernel"] = rng.normal(size=(4096, 10)).astype(np.float32)
    P2["params"]["linear"]["bias"] = rng.normal(size=(10,)).astype(np.float32)
    Ps = configs.synth_params(2)
    Ps["params"]["linear"]["kernel"] = P["params"]["linear"]["kernel"] + P2["params"]["linear"]["kernel"]
    Ps["params"]["linear"]["bias"] = P["params"]["linear"]["bias"] + P2["params"]["linear"]["bias"]
    o1, o2, os_ = out.cpu().numpy(), net.apply(P2, xt).cpu().numpy(), net.apply(Ps, xt).cpu().numpy()
    assert np.abs(os_ - (o1 + o2)).max() <= 2e-5 * np.abs(os_).max()


def test_cfg5_imq_16384_centres_subset(gpu):
    cfg = configs.model_card(5)
    P = configs.synth_params(5)
    x = configs.synth_queries(5, B=8192)
    out = WCRBFNet.from_config(cfg).apply(P, x)
    exp = co.wcrbf_forward(cfg, P, x[:512], np.float64)
    assert relmax(out[:512], exp) <= RTOL


# ------------------------------------------------------------------ roll-outs
def test_kat1_on_gpu(gpu, kat):
    k = kat["kat1"]
    xu = np.hstack([np.zeros((10, 7), np.float32), np.full((10, 10), k["u"], np.float32)])
    got = dyn.integrate_st_mult(xu, k["dyn_params"])
    exp = np.array(k["all_states_row"], np.float32)
    assert got.shape == (10, 5, 7)
    np.testing.assert_allclose(got, np.broadcast_to(exp, got.shape), rtol=1e-6, atol=1e-7)


def _st_inputs(B, T, seed, fast=True):
    rng = np.random.default_rng(seed)
    st = rng.normal(size=(B, 7)) * [2, 2, .3, 1, 1, .4, .1]
    st[:, 3] = rng.uniform(0.2, 7.5, B) if fast else rng.uniform(0.0, 2.9, B)
    u = np.hstack([rng.normal(size=(B, T)) * 5.0, rng.normal(size=(B, T)) * 2.0])
    return np.hstack([st, u])


def _frenet_inputs(B, T, rng):
    """[s, ey, delta, vx, vy, wz, epsi, cur] + controls; curvature / offsets kept in the range of a
    race track (|ey*cur| << 1) so that 1/(1 - ey*cur) (dynamics.py:268) stays away from its pole over T steps."""
    st = rng.normal(size=(B, 8)) * [1, .2, .2, 1, .1, .1, .15, .08] + [0, 0, 0, 4, 0, 0, 0, 0]
    amp = 1.0 if T <= 10 else 0.25
    return np.hstack([st, rng.normal(size=(B, T)) * 5 * amp, rng.normal(size=(B, T)) * 2 * amp])


@pytest.mark.parametrize("T", [1, 5, 50])
@pytest.mark.parametrize("B", [1, 64, 1000])
def test_rollouts_match_oracle(gpu, B, T):
    xu = _st_inputs(B, T, seed=B + T)
    xu32 = xu.astype(F32)
    assert_states_close(dyn.integrate_st_mult(xu32, DP), orc.integrate_st_mult(xu32.astype(np.float64), DP),
                        orc.integrate_st_mult(xu32, DP32))
    assert_states_close(dyn.integrate_st_ks_mult(xu32, DP), orc.integrate_st_ks_mult(xu32.astype(np.float64), DP),
                        orc.integrate_st_ks_mult(xu32, DP32))
    rng = np.random.default_rng(B * 7 + T)
    xf = _frenet_inputs(B, T, rng).astype(F32)
    assert_states_close(dyn.integrate_frenet_mult(xf, DP), orc.integrate_frenet_mult(xf.astype(np.float64), DP),
                        orc.integrate_frenet_mult(xf, DP32))
    v0 = rng.uniform(-1, 8, B).astype(F32)
    u = np.hstack([rng.normal(size=(B, T)) * 5, rng.normal(size=(B, T)) * 2]).astype(F32)
    assert_states_close(dyn.rollout_fullint(v0, u), orc.rollout_fullint(v0.astype(np.float64), u.astype(np.float64)),
                        orc.rollout_fullint(v0, u))


def test_onestep_aux_and_spiral(gpu):
    rng = np.random.default_rng(4)
    xu = _st_inputs(333, 1, seed=4).astype(F32)
    np.testing.assert_allclose(dyn.dynamic_st_onestep_aux(xu, DP),
                               orc.dynamic_st_onestep_aux(xu.astype(np.float64), DP), rtol=RTOL, atol=2e-6)
    q = np.hstack([rng.normal(size=(500, 4)) * .3, rng.uniform(1, 10, size=(500, 1))])
    q32 = q.astype(F32)
    assert_states_close(pu.integrate_path_mult(q32), orc.integrate_path_mult(q32.astype(np.float64)),
                        orc.integrate_path_mult(q32))
    st = pu.integrate_path_mult(np.array([[0, 0, 0, 0, 7.5]], np.float32))
    np.testing.assert_allclose(st[0, -1, :3], [7.5, 0, 0], atol=1e-6)        # straight line (App. A.6)
    with np.errstate(all="ignore"):
        assert np.isnan(pu.integrate_path_mult(np.array([[.1, .2, .1, 0, 0]], np.float32))).any()   # s = 0 (B-8)


def test_rollout_full_size_mirror_symmetry(gpu):
    """BASELINE cfg-4 size on one GPU: B = 32768 trajectories, T = 50; mirror symmetry (App. A.6)."""
    torch = gpu
    B, T = 32768, 50
    xu = _st_inputs(B, T, seed=99).astype(np.float32)
    m = xu.copy()
    m[:, [1, 2, 4]] *= -1
    m[:, 7 + T:] *= -1
    a = dyn.integrate_st_ks_mult(torch.from_numpy(xu).cuda(), DP)
    b = dyn.integrate_st_ks_mult(torch.from_numpy(m).cuda(), DP)
    sign = torch.tensor([1, -1, -1, 1, -1, 1, 1.0], device="cuda")
    assert torch.allclose(a, b * sign, rtol=1e-6, atol=1e-6)
    sub = np.arange(0, B, 97)
    assert_states_close(a.cpu().numpy()[sub], orc.integrate_st_ks_mult(xu[sub].astype(np.float64), DP),
                        orc.integrate_st_ks_mult(xu[sub], DP32))


@pytest.mark.parametrize("T", [3, 7, 9, 33])
def test_rollout_odd_horizons_and_misaligned_buffers(gpu, T):
    """The flush / fetch paths cut every row into aligned 16-byte pieces from the REAL addresses: feed
    them views whose base pointer is only 4-byte aligned, batches that are not multiples of 64 and
    horizons that are not multiples of the 4-step chunk."""
    torch = gpu
    for B, off in ((1, 1), (65, 3), (130, 2), (257, 1)):
        xu = _st_inputs(B, T, seed=B + T).astype(F32)
        big = torch.zeros(off + xu.size, dtype=torch.float32, device="cuda")
        view = big[off:off + xu.size].view(B, -1)                # data_ptr is base + 4*off bytes
        view.copy_(torch.from_numpy(xu))
        assert view.data_ptr() % 16 == (4 * off) % 16 and view.is_contiguous()
        got = dyn.integrate_st_ks_mult(view, DP).cpu().numpy()
        ref = dyn.integrate_st_ks_mult(torch.from_numpy(xu).cuda(), DP).cpu().numpy()
        np.testing.assert_array_equal(got, ref)                  # same arithmetic whatever the alignment
        assert_states_close(got, orc.integrate_st_ks_mult(xu.astype(np.float64), DP), orc.integrate_st_ks_mult(xu, DP32))
        got_sel = dyn.integrate_st_mult(view, DP).cpu().numpy()
        assert_states_close(got_sel, orc.integrate_st_mult(xu.astype(np.float64), DP), orc.integrate_st_mult(xu, DP32))


def test_hip_graph_capture_and_replay(gpu):
    """No hipMalloc / synchronisation on the launch path: a forward + roll-out sequence can be captured
    into a HIP graph and replayed (include/irbfn_hip.h conventions)."""
    torch = gpu
    cfg = configs.model_card(2)
    net = WCRBFNet.from_config(cfg)
    P = configs.synth_params(2)
    x = torch.from_numpy(configs.synth_queries(2, B=4096)).cuda()
    xu = torch.from_numpy(_st_inputs(4096, 5, seed=3).astype(F32)).cuda()
    net.bind(P)
    ref_out, ref_st = net(x), dyn.integrate_st_mult(xu, DP)     # warm-up: descriptors, attributes
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            net(x)
            dyn.integrate_st_mult(xu, DP)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(x)
        st = dyn.integrate_st_mult(xu, DP)
    out.zero_()
    st.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out) and torch.equal(st, ref_st)


def test_rollout_input_validation(gpu):
    with pytest.raises(ValueError):
        dyn.integrate_st_mult(np.zeros((4, 16), np.float32), DP)
    with pytest.raises(ValueError):
        dyn.dynamic_st_onestep_aux(np.zeros((4, 17), np.float32), DP)
    with pytest.raises(ValueError):
        dyn.integrate_st_mult(np.zeros((4, 17), np.float32), DP[:5])
    assert dyn.integrate_st_mult(np.zeros((0, 17), np.float32), DP).shape == (0, 5, 7)


# ------------------------------------------------------------------ fused planning tick
@pytest.mark.parametrize("mode,T", [(_lib.ROLLOUT_ST_SELECT, 5), (_lib.ROLLOUT_ST_KS, 5), (_lib.ROLLOUT_ST_KS, 50)])
def test_fused_forward_rollout_equals_two_launches(gpu, mode, T):
    torch = gpu
    from irbfn_amd.planner import plan_batch
    cfg = dict(configs.model_card(2), out_features=2 * T, num_kernels=512)
    rng = np.random.default_rng(T)
    P = {"params": {"rbf_list": {"centers": rng.uniform(-1, 8, size=(1, 512, 7)).astype(np.float32),
                                 "log_sigs": rng.uniform(0, 2, size=(1, 512)).astype(np.float32)},
                    "linear": {"kernel": (rng.normal(size=(512, 2 * T)) * 0.3).astype(np.float32),
                               "bias": np.zeros(2 * T, np.float32)}}}
    net = WCRBFNet.from_config(cfg)
    x = configs.synth_queries(2, B=777)
    st0 = configs.initial_state_from_query(x)
    u, states = plan_batch(net, P, x, st0, DP, mode=mode)
    u2 = net.apply(P, x)                       # B = 777 takes the tiled centre-lane kernel: other summation order
    assert np.abs(u - u2).max() <= 2e-6 * np.abs(u2).max()
    fn = dyn.integrate_st_mult if mode == _lib.ROLLOUT_ST_SELECT else dyn.integrate_st_ks_mult
    np.testing.assert_array_equal(states, fn(np.hstack([st0, u]), DP))      # same step function, no FP contraction


# ------------------------------------------------------------------ VJPs
@pytest.mark.parametrize("run", ["dnmpc_1regions_newdata_oldintloss_nomirror_highk", "dnmpc_128regions",
                                 "dnmpc_12regions_frenet_l1_bigdata"])
def test_net_vjp_trained_checkpoints(gpu, run):
    cfg, params, x, *_ = load_ckpt_fixture(run)
    g = np.random.default_rng(5).normal(size=(x.shape[0], cfg["out_features"]))
    ref = orc.wcrbfnet_vjp(cfg, params, x, g)["params"]
    net = WCRBFNet.from_config(cfg)
    got = net.vjp(_p32(params), x.astype(np.float32), g.astype(np.float32))["params"]
    for grp, name in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
        a, b = got[grp][name], ref[grp][name]
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 5e-5 * np.abs(b).max() + 1e-7, (name, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("basis", ["gaussian", "gaussian_wide", "inverse_quadratic", "inverse_multiquadric",
                                   "multiquadric", "quadratic"])
def test_net_vjp_bases_and_ragged(gpu, basis):
    cfg = dict(configs.model_card(1), basis_func=basis)
    P = configs.synth_params(1)
    for B in (1, 100, 1500):
        x = configs.synth_queries(1, B=B)
        g = configs.synth_cotangent(1, B=B)
        ref = orc.wcrbfnet_vjp(cfg, P, x, g)["params"]
        got = WCRBFNet.from_config(cfg).vjp(P, x, g)["params"]
        for grp, name in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
            a, b = got[grp][name], ref[grp][name]
            assert np.abs(a - b).max() <= 5e-5 * np.abs(b).max() + 1e-7, (basis, B, name)


@pytest.mark.parametrize("basis", ["linear", "spline", "poisson_one", "poisson_two", "matern32", "matern52"])
def test_net_vjp_distance_dependent_bases(gpu, basis):
    """The bases that depend on d = sqrt(d^2) itself (flax_rbf.py:55-111; no reference model card uses them): parameter
    VJP against torch.autograd of the float64 restatement, two regions and ragged batches; a query ON a centre gives
    NaN in that centre's gradient, as jax.grad of the reference does (sqrt has no derivative at 0)."""
    torch = gpu
    cfg = dict(configs.model_card(1), basis_func=basis)
    P = configs.synth_params(1)
    net = WCRBFNet.from_config(cfg)
    for B in (1, 100, 1500):
        x = configs.synth_queries(1, B=B)
        g = configs.synth_cotangent(1, B=B)
        tp = orc.torch_params(orc.cast_params(P, np.float64), torch.float64, requires_grad=True)
        out = orc.wcrbfnet_apply(cfg, tp, torch.tensor(x, dtype=torch.float64))
        (out * torch.tensor(g, dtype=torch.float64)).sum().backward()
        got = net.vjp(P, x, g)["params"]
        for grp, name in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
            a, b = got[grp][name], tp["params"][grp][name].grad.numpy()
            assert np.abs(a - b).max() <= 5e-5 * np.abs(b).max() + 1e-7, (basis, B, name, np.abs(a - b).max() / np.abs(b).max())
    x = configs.synth_queries(1, B=64).copy()
    c = np.asarray(P["params"]["rbf_list"]["centers"], np.float32)
    x[7] = c[0, 5]                                                          # query 7 sits on centre 5 of region 0
    got = net.vjp(P, x, configs.synth_cotangent(1, B=64))["params"]
    gc = got["rbf_list"]["centers"]
    assert np.isnan(gc[0, 5]).all()
    mask = np.ones(gc.shape[:2], bool); mask[0, 5] = False
    assert np.isfinite(gc[mask]).all() and np.isfinite(got["linear"]["kernel"]).all()
    # the WIDTH does not go through the sqrt (flax_rbf.py:280: sqrt(sum sq) / exp(log_sig)): d phi / d log_sig =
    # phi'(d) * (-d) = 0 at d = 0 -- finite everywhere, and equal to torch.autograd of the restatement, which shows
    # the same split (NaN on the centre path only)
    xg = configs.synth_cotangent(1, B=64)
    tp = orc.torch_params(orc.cast_params(P, np.float64), torch.float64, requires_grad=True)
    out = orc.wcrbfnet_apply(cfg, tp, torch.tensor(x, dtype=torch.float64))
    (out * torch.tensor(xg, dtype=torch.float64)).sum().backward()
    ref_ls = tp["params"]["rbf_list"]["log_sigs"].grad.numpy()
    ref_c = tp["params"]["rbf_list"]["centers"].grad.numpy()
    assert np.isfinite(ref_ls).all() and np.isnan(ref_c[0, 5]).all()
    gls = got["rbf_list"]["log_sigs"]
    assert np.isfinite(gls).all()
    assert np.abs(gls - ref_ls).max() <= 5e-5 * np.abs(ref_ls).max() + 1e-7


def test_net_vjp_cfg3_size_determinism_and_subset(gpu):
    torch = gpu
    cfg = configs.model_card(3)
    P = configs.synth_params(3)
    net = WCRBFNet.from_config(cfg)
    B = 16384
    x, g = configs.synth_queries(3, B=B), configs.synth_cotangent(3, B=B)
    xt, gt = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    a = net.vjp(P, xt, gt)["params"]
    b = net.vjp(P, xt, gt)["params"]
    for grp, name in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
        assert torch.equal(a[grp][name], b[grp][name])          # no float atomics: bitwise reproducible
    # linearity in the cotangent: vjp(g1 + g2) = vjp(g1) + vjp(g2)
    g2 = torch.from_numpy(configs.synth_cotangent(3, B=B, seed=77)).cuda()
    c = net.vjp(P, xt, g2)["params"]
    s = net.vjp(P, xt, gt + g2)["params"]
    for grp, name in (("rbf_list", "centers"), ("linear", "kernel")):
        lhs, rhs = s[grp][name], a[grp][name] + c[grp][name]
        assert (lhs - rhs).abs().max() <= 1e-4 * rhs.abs().max()
    # parity of d kernel / d bias against the oracle on the full batch (cheap: h^T g)
    _, h, _ = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x[:2048].astype(np.float64), return_aux=True, chunk=256)
    a2 = net.vjp(P, xt[:2048].contiguous(), gt[:2048].contiguous())["params"]
    refk = h.T @ g[:2048].astype(np.float64)
    assert relmax(a2["linear"]["kernel"].cpu().numpy(), refk) <= 5e-5
    np.testing.assert_allclose(a2["linear"]["bias"].cpu().numpy(), g[:2048].astype(np.float64).sum(0), rtol=1e-4, atol=1e-4)


def test_rollout_vjps_match_hand_adjoints(gpu):
    rng = np.random.default_rng(21)
    for B, T in ((1, 1), (70, 5), (300, 50)):
        xu = _st_inputs(B, T, seed=B)
        gs = rng.normal(size=(B, T, 7))
        ref = hv.vjp_st_ks(xu, DP, gs)
        got = dyn.rollout_vjp(_lib.ROLLOUT_ST_KS, xu.astype(np.float32), DP, gs.astype(np.float32), T)
        assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max() + 1e-6
        v0, u = rng.uniform(-1, 8, B), np.hstack([rng.normal(size=(B, T)) * 5, rng.normal(size=(B, T)) * 2])
        g5 = rng.normal(size=(B, T, 5))
        gv, gu = hv.vjp_fullint(v0, u, g5)
        got = dyn.rollout_vjp(_lib.ROLLOUT_FULLINT, np.hstack([v0[:, None], u]).astype(np.float32), None,
                              g5.astype(np.float32), T)
        ref = np.hstack([gv[:, None], gu])
        assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max() + 1e-6
        xf = _frenet_inputs(B, T, rng).astype(F32).astype(np.float64)
        g8 = rng.normal(size=(B, T, 8))
        ref = hv.vjp_frenet(xf, DP, g8)
        got = dyn.rollout_vjp(_lib.ROLLOUT_FRENET_LS, xf.astype(np.float32), DP, g8.astype(np.float32), T)
        rowmax = np.abs(ref).max(axis=1, keepdims=True)
        assert (np.abs(got - ref) <= 1e-4 * rowmax + 1e-6).all()
    q = np.hstack([rng.normal(size=(200, 4)) * .3, rng.uniform(1, 10, size=(200, 1))])
    g6 = rng.normal(size=(200, 9, 6))
    ref = hv.vjp_spiral(q, g6)
    got = dyn.rollout_vjp(_lib.ROLLOUT_SPIRAL, q.astype(np.float32), None, g6.astype(np.float32), 9)
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()
    with pytest.raises(ValueError):      # ST_SELECT is never differentiated by the reference (App. B-5)
        dyn.rollout_vjp(_lib.ROLLOUT_ST_SELECT, np.zeros((4, 17), np.float32), DP, np.zeros((4, 5, 7), np.float32), 5)


def test_clip_tie_rule(gpu):
    """Inputs sitting exactly on a bound (training grids do: v = v_max, App. B-7)."""
    T = 3
    xu = _st_inputs(8, T, seed=1)
    xu[:, 3] = 7.0                       # V == v_max
    xu[:, 7:7 + T] = 9.51                # a == a_max
    gs = np.random.default_rng(3).normal(size=(8, T, 7))
    for tie in (0.0, 0.5, 1.0):
        ref = hv.vjp_st_ks(xu, DP, gs, tie)
        got = dyn.rollout_vjp(_lib.ROLLOUT_ST_KS, xu.astype(np.float32), DP, gs.astype(np.float32), T, clip_tie=tie)
        assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max() + 1e-6


def test_autograd_train_step_matches_oracle_grad(gpu):
    """value_and_grad of train_step_oneint's loss (scripts/train_nmpc.py:268-298) through the HIP path."""
    import torch
    from irbfn_amd import autograd as ag
    cfg, params, x, *_ = load_ckpt_fixture("dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq")
    y = np.random.default_rng(2).normal(size=(64, 2))
    tp = orc.torch_params(params, torch.float64, requires_grad=True)
    loss_ref = orc.train_oneint_loss(cfg, tp, torch.tensor(x), torch.tensor(y), DP)
    loss_ref.backward()
    net = WCRBFNet.from_config(cfg)
    P = {"params": {"rbf_list": {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda", requires_grad=True)
                                 for k, v in params["params"]["rbf_list"].items()},
                    "linear": {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda", requires_grad=True)
                               for k, v in params["params"]["linear"].items()}}}
    loss = ag.train_oneint_loss(net, P, torch.tensor(x, dtype=torch.float32, device="cuda"),
                                torch.tensor(y, dtype=torch.float32, device="cuda"), DP)
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))
    for grp, name in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
        a = P["params"][grp][name].grad.cpu().numpy()
        b = tp["params"][grp][name].grad.numpy()
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max() + 1e-9, name


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 64, 333, 4096])
def test_deeper_wcrbfnet_forward(gpu, B):
    """DeeperWCRBFNet (model.py:201-289; IRBFNFrenetPlanner(deeper=True)) on the reference's
    trained checkpoint: fused RBF + linear_pre1 kernel, then the Dense head kernel, vs the float64 oracle."""
    from irbfn_amd.model import DeeperWCRBFNet
    cfg, params, x, out64 = load_deeper_fixture()
    net = DeeperWCRBFNet.from_config(cfg)
    if B <= x.shape[0]:
        xs, ref = x[:B], out64[:B]
    else:
        ns = len(cfg["activation_idx"])
        lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
        xs = np.random.default_rng(B).uniform(lo, hi, size=(B, cfg["in_features"])).astype(np.float32).astype(np.float64)
        p64 = {"params": {k: {n: np.asarray(v, np.float64) for n, v in d.items()} for k, d in params["params"].items()}}
        ref = orc.deeper_wcrbfnet_apply(cfg, p64, xs)
    got = net.apply(params, xs.astype(np.float32))
    assert got.shape == (B, cfg["out_features"]) and got.dtype == np.float32
    scale = np.abs(ref).max()
    # float32 path vs float64 oracle on float32-rounded x: 2e-5 of the output scale, or (the trained net's
    # hidden layer cancels heavily) 4x the error the float32 NumPy restatement itself makes on these rows
    p32 = {"params": {k: {n: np.asarray(v, np.float32) for n, v in d.items()} for k, d in params["params"].items()}}
    err32 = np.abs(orc.deeper_wcrbfnet_apply(cfg, p32, xs.astype(np.float32)).astype(np.float64) - ref).max()
    err = np.abs(got - ref).max()
    assert err <= max(2e-5 * scale + 1e-5, 4 * err32), (err, err32, scale)
    # params with a wrong head shape are rejected before any launch
    bad = {"params": dict(params["params"], linear_pre2={"kernel": np.zeros((64, 32)), "bias": np.zeros(32)})}
    with pytest.raises(ValueError):
        net.apply(bad, xs.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("R,K,O,B", [(11, 20, 10, 500), (2, 64, 2, 70), (1, 33, 5, 130), (64, 8, 10, 3000)])
def test_cluster_wcrbfnet_forward(gpu, R, K, O, B):
    """ClusterWCRBFNet (model.py:341-414): softmax gate kernel + fused forward with external region
    weights, against the float64 restatement.  No trained checkpoint of this variant survives in the reference."""
    from irbfn_amd.model import ClusterWCRBFNet
    rng = np.random.default_rng(R * 7 + K)
    D = 8
    cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": "gaussian", "num_regions": R}
    params = {"params": {
        "rbf_list": {"centers": rng.uniform(-2, 2, size=(R, K, D)).astype(np.float32),
                     "log_sigs": rng.uniform(0.0, 1.0, size=(R, K)).astype(np.float32)},
        "linear": {"kernel": rng.normal(size=(K, O)).astype(np.float32), "bias": rng.normal(size=(O,)).astype(np.float32)},
        "cluster": {"kernel": rng.normal(size=(D, R)).astype(np.float32) * 2.0, "bias": rng.normal(size=(R,)).astype(np.float32)}}}
    x = rng.uniform(-2, 2, size=(B, D)).astype(np.float32)
    net = ClusterWCRBFNet(**cfg)
    out, logits = net.apply(params, x)
    p64 = {"params": {k: {n: np.asarray(v, np.float64) for n, v in d.items()} for k, d in params["params"].items()}}
    ref_out, ref_logits = orc.cluster_wcrbfnet_apply(cfg, p64, x.astype(np.float64))
    assert out.shape == (B, O) and logits.shape == (B, R)
    assert np.abs(logits - ref_logits).max() <= 1e-5 * (1 + np.abs(ref_logits).max())
    assert np.abs(out - ref_out).max() <= 2e-5 * np.abs(ref_out).max() + 1e-5
    bad = {"params": dict(params["params"], cluster={"kernel": np.zeros((D, R + 1), np.float32), "bias": np.zeros(R + 1, np.float32)})}
    with pytest.raises(ValueError):
        net.apply(bad, x)


def test_bind_tracks_numpy_leaves_by_content(gpu):
    """apply(params, x) with NumPy leaves: an in-place change is seen, FRESH arrays with new contents are seen even
    when CPython hands them the ids / buffers of the dropped ones (the pattern of the documented pure_callback
    adapter), and unchanged contents skip the upload.  torch leaves are tracked by object + version."""
    import torch
    cfg, P, x = configs.model_card(1), configs.synth_params(1), configs.synth_queries(1, B=100)
    net = WCRBFNet.from_config(cfg)
    a = net.apply(P, x)
    P["params"]["linear"]["bias"] += 1.0                        # in-place change of a writable leaf must be seen
    b = net.apply(P, x)
    assert np.abs((b - a) - 1.0).max() < 1e-5
    dev = torch.cuda.current_device()
    fp = net._bound_fp[dev]
    c = net.apply(P, x)                                         # same contents: no re-upload, same result
    assert net._bound_fp[dev] is fp and np.array_equal(c, b)
    base = b
    for step in range(1, 12):                                   # fresh frozen temporaries per step, then dropped
        tmp = {"params": {g: {n: np.array(v) for n, v in d.items()} for g, d in P["params"].items()}}
        tmp["params"]["linear"]["bias"] = tmp["params"]["linear"]["bias"] + float(step)
        for d in tmp["params"].values():
            for v in d.values():
                v.setflags(write=False)
        got = net.apply(tmp, x)
        assert np.abs((got - base) - float(step)).max() < 1e-4, step
        del tmp
    # torch leaves: in-place update bumps the version -> re-upload; an untouched tensor pytree is skipped
    Pt = {"params": {g: {n: torch.from_numpy(np.array(v)).cuda() for n, v in d.items()} for g, d in P["params"].items()}}
    t0 = net.apply(Pt, x)
    fpt = net._bound_fp[dev]
    net.apply(Pt, x)
    assert net._bound_fp[dev] is fpt
    Pt["params"]["linear"]["bias"].add_(2.0)
    t1 = net.apply(Pt, x)
    assert np.abs((t1 - t0) - 2.0).max() < 1e-5


