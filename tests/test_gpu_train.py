"""GPU parity of the on-device training step against the oracle: loss value, seeds /
parameter gradients (torch.autograd of the restatement = the reference's jax.value_and_grad) and
the optax-style clip + Adam update, for train_step_oneint and train_step_fullint."""
import numpy as np
import pytest
import torch

from conftest import load_ckpt_fixture
from irbfn_amd import configs, distributed, train
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

pytestmark = pytest.mark.gpu
DP = np.array(configs.DYN_PARAMS)
LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


def _flat(p):
    p = p["params"]
    return np.concatenate([np.asarray(p[g][n], np.float64).reshape(-1) for g, n in LEAVES])


def _oracle_step(cfg, params, x, y, loss_fn, lr, max_norm, m, v, t):
    tp = orc.torch_params(params, torch.float64, requires_grad=True)
    loss = loss_fn(tp)
    loss.backward()
    g = np.concatenate([tp["params"][g_][n].grad.numpy().reshape(-1) for g_, n in LEAVES])
    gc = orc.clip_by_global_norm(g, max_norm)
    p_new, m, v = orc.adam_update(_flat(params), gc, m, v, t, lr=lr)
    return float(loss), g, p_new, m, v


def _unflat(net, flat):
    return {"params": {k: {n: t.numpy() for n, t in d.items()}
                       for k, d in distributed.unflatten_params(net, torch.from_numpy(flat))["params"].items()}}


@pytest.mark.parametrize("max_norm", [1.0, 1e-3])
def test_train_step_oneint_two_steps(gpu, max_norm):
    cfg, params, x, *_ = load_ckpt_fixture("dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq")
    params = orc.cast_params(params, np.float32)
    y = np.random.default_rng(2).normal(size=(64, 2)).astype(np.float32)
    x32 = x.astype(np.float32)
    net = WCRBFNet.from_config(cfg)
    state = train.TrainState.create(net, params, lr=1e-3, max_grad_norm=max_norm)
    n = state.flat.numel()
    m, v = np.zeros(n), np.zeros(n)
    cur = orc.cast_params(params, np.float64)
    for t in (1, 2):
        loss_ref, g_ref, p_ref, m, v = _oracle_step(
            cfg, cur, x32, y, lambda tp: orc.train_oneint_loss(cfg, tp, torch.tensor(x32, dtype=torch.float64),
                                                               torch.tensor(y, dtype=torch.float64), DP),
            1e-3, max_norm, m, v, t)
        state, loss = train.train_step_oneint(state, torch.from_numpy(x32).cuda(), torch.from_numpy(y).cuda(), DP)
        assert abs(float(loss) - loss_ref) <= 3e-5 * abs(loss_ref)
        g_gpu = state.g.cpu().numpy()
        assert np.abs(g_gpu - g_ref).max() <= 2e-4 * np.abs(g_ref).max()
        p_gpu = state.flat.cpu().numpy()
        # Adam's first steps move every parameter by ~lr regardless of the gradient scale
        assert np.abs(p_gpu - p_ref).max() <= 2e-5 + 1e-6 * np.abs(p_ref).max()
        assert int(state.step.item()) == t
        cur = _unflat(net, p_ref)
    # the updated parameters are what the next forward uses
    out = net.apply(state.params, torch.from_numpy(x32).cuda()).cpu().numpy()
    ref = orc.wcrbfnet_apply(cfg, cur, x32.astype(np.float64))
    assert np.abs(out - ref).max() <= 2e-3 * np.abs(ref).max()


def test_train_step_fullint(gpu):
    rng = np.random.default_rng(5)
    cfg = dict(configs.model_card(2), num_kernels=300, out_features=10)
    params = {"params": {"rbf_list": {"centers": rng.uniform(-1, 8, size=(1, 300, 7)).astype(np.float32),
                                      "log_sigs": rng.uniform(0, 2, size=(1, 300)).astype(np.float32)},
                         "linear": {"kernel": (rng.normal(size=(300, 10)) * 0.3).astype(np.float32),
                                    "bias": np.zeros(10, np.float32)}}}
    x = configs.synth_queries(2, B=500)
    y = np.hstack([rng.normal(size=(500, 5)) * 3, rng.normal(size=(500, 5))]).astype(np.float32)
    net = WCRBFNet.from_config(cfg)
    state = train.TrainState.create(net, params, lr=1e-3, max_grad_norm=1.0)
    n = state.flat.numel()
    loss_ref, g_ref, p_ref, _, _ = _oracle_step(
        cfg, orc.cast_params(params, np.float64), x, y,
        lambda tp: orc.train_fullint_loss(cfg, tp, torch.tensor(x, dtype=torch.float64), torch.tensor(y, dtype=torch.float64)),
        1e-3, 1.0, np.zeros(n), np.zeros(n), 1)
    state, loss = train.train_step_fullint(state, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    assert abs(float(loss) - loss_ref) <= 3e-5 * abs(loss_ref)
    g_gpu = state.g.cpu().numpy()
    assert np.abs(g_gpu - g_ref).max() <= 5e-4 * np.abs(g_ref).max()
    assert np.abs(state.flat.cpu().numpy() - p_ref).max() <= 5e-5 + 1e-6 * np.abs(p_ref).max()


def test_train_step_frenet_fullint(gpu):
    """Frenet train_step_fullint (scripts/train_nmpc_frenet.py:394-421) on the reference's trained 12-region Frenet
    checkpoint: loss, parameter gradient (through integrate_frenet_mult) and the Adam update against
    torch.autograd of the restatement."""
    cfg, params, x0, *_ = load_ckpt_fixture("dnmpc_12regions_frenet_l1_bigdata")
    rng = np.random.default_rng(9)
    T = 5
    cfg = dict(cfg, out_features=2 * T)                      # the fixture net has O = 2: widen the Dense to 2T = 10
    K = cfg["num_kernels"]
    params = orc.cast_params(params, np.float32)
    params["params"]["linear"] = {"kernel": (rng.normal(size=(K, 2 * T)) * 0.05).astype(np.float32),
                                  "bias": (rng.normal(size=(2 * T,)) * 0.1).astype(np.float32)}
    B = 300
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = rng.uniform(lo, hi, size=(B, 8)).astype(np.float32)
    x[:, 7] = rng.normal(size=B).astype(np.float32) * 0.05   # curvature of a race track: 1 - ey * cur stays away from 0
    x[:, 0] = rng.normal(size=B).astype(np.float32) * 0.2
    y = np.hstack([rng.normal(size=(B, T)) * 2, rng.normal(size=(B, T)) * 0.5]).astype(np.float32)
    net = WCRBFNet.from_config(cfg)
    state = train.TrainState.create(net, params, lr=1e-3, max_grad_norm=1.0)
    n = state.flat.numel()
    loss_ref, g_ref, p_ref, _, _ = _oracle_step(
        cfg, orc.cast_params(params, np.float64), x, y,
        lambda tp: orc.train_frenet_fullint_loss(cfg, tp, torch.tensor(x, dtype=torch.float64),
                                                 torch.tensor(y, dtype=torch.float64), DP),
        1e-3, 1.0, np.zeros(n), np.zeros(n), 1)
    state, loss = train.train_step_frenet_fullint(state, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), DP)
    assert abs(float(loss) - loss_ref) <= 3e-5 * abs(loss_ref), (float(loss), loss_ref)
    g_gpu = state.g.cpu().numpy()
    assert np.abs(g_gpu - g_ref).max() <= 5e-4 * np.abs(g_ref).max(), np.abs(g_gpu - g_ref).max() / np.abs(g_ref).max()
    assert np.abs(state.flat.cpu().numpy() - p_ref).max() <= 5e-5 + 1e-6 * np.abs(p_ref).max()


def test_training_reduces_the_loss(gpu):
    cfg = dict(configs.model_card(2), num_kernels=256, out_features=2)
    net = WCRBFNet.from_config(cfg)
    rng = np.random.default_rng(0)
    params = net.init(seed=3)
    params["params"]["rbf_list"]["centers"] = rng.uniform(-1, 8, size=(1, 256, 7)).astype(np.float32)
    x = torch.from_numpy(configs.synth_queries(2, B=4096)).cuda()
    y = torch.stack([torch.sin(x[:, 1]) * 3.0, torch.cos(x[:, 3])], dim=1).contiguous()
    state = train.TrainState.create(net, params, lr=1e-2, max_grad_norm=1.0)
    losses = []
    for _ in range(60):
        state, loss = train.train_step_oneint(state, x, y, DP)
        losses.append(loss)
    l = torch.cat(losses).cpu().numpy()
    assert np.isfinite(l).all() and l[-1] < 0.6 * l[0]


def test_train_epoch_over_device_table(gpu, tmp_path):
    """.npz table -> prepare (mirror + flatten) -> model card -> device-resident epochs
    (scripts/train_nmpc.py:455-486).  Every row is visited at most once per epoch, the remainder is
    dropped, the loss decreases over epochs."""
    from irbfn_amd import tables
    rng = np.random.default_rng(5)
    n = 3000
    inputs = rng.uniform([0, 0, 0, 0, 0, -0.6, -3.0], [7, 3.6, 3.6, 3.2, 7, 0.4, 2.5], size=(n, 7)).round(2)
    outputs = np.stack([np.sin(inputs[:, [1]]) * np.linspace(1, 2, 5), np.cos(inputs[:, [2]]) * np.linspace(0.2, 1, 5)], axis=-1)
    path = str(tmp_path / "table.npz")
    np.savez(path, inputs=inputs, outputs=outputs)
    fx, fy = tables.prepare(path, tables.CARTESIAN, mirror_data=True)
    assert fx.shape == (2 * n, 7) and fy.shape == (2 * n, 10)
    card = tables.model_card(fx, fy, [1] * 7, 128, "gaussian")
    net = WCRBFNet.from_config(card)
    params = net.init(seed=1)
    params["params"]["rbf_list"]["centers"] = rng.uniform(-4, 8, size=(1, 128, 7)).astype(np.float32)
    table = tables.DeviceTable(fx, fy, seed=7)
    batches = list(table.epoch(1024))
    assert len(batches) == 5 and batches[0][0].shape == (1024, 7) and batches[0][1].shape == (1024, 10)
    seen = torch.cat([b[0] for b in batches])
    # rows of one epoch are distinct table rows (x rows are unique with overwhelming probability)
    assert torch.unique(seen, dim=0).shape[0] == seen.shape[0]
    state = train.TrainState.create(net, params, lr=5e-3, max_grad_norm=1.0)
    means = []
    for _ in range(8):
        state, losses = train.train_epoch(state, table, 1024)
        assert losses.shape == (5,)
        means.append(float(losses.mean()))
    assert np.isfinite(means).all() and means[-1] < 0.95 * means[0], means


def test_deeper_wcrbfnet_vjp_on_the_reference_checkpoint(gpu):
    """DeeperWCRBFNet (model.py:201-289) on the reference's trained `..._5stepint_deeper` checkpoint: all eight
    gradient leaves of a random cotangent against torch.autograd of the restatement; deterministic."""
    from conftest import load_deeper_fixture
    from irbfn_amd.model import DeeperWCRBFNet
    cfg, params, x, out64 = load_deeper_fixture()
    rng = np.random.default_rng(3)
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    B = 700
    xx = rng.uniform(lo, hi, size=(B, cfg["in_features"])).astype(np.float32)
    xx[:64] = x.astype(np.float32)
    O = cfg["out_features"]
    g = rng.normal(size=(B, O)).astype(np.float32)
    p32 = {"params": {k: {n: np.asarray(v, np.float32) for n, v in d.items()} for k, d in params["params"].items()}}
    net = DeeperWCRBFNet.from_config(cfg)
    a = net.vjp(p32, torch.from_numpy(xx).cuda(), torch.from_numpy(g).cuda())["params"]
    a2 = net.vjp(p32, torch.from_numpy(xx).cuda(), torch.from_numpy(g).cuda())["params"]
    tp = {"params": {k: {n: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for n, v in d.items()}
                     for k, d in p32["params"].items()}}
    out = orc.deeper_wcrbfnet_apply(cfg, tp, torch.tensor(xx, dtype=torch.float64))
    (out * torch.tensor(g, dtype=torch.float64)).sum().backward()
    for grp in ("rbf_list", "linear_pre1", "linear_pre2", "linear"):
        for name, t in tp["params"][grp].items():
            ref = t.grad.numpy()
            got = a[grp][name].cpu().numpy()
            assert torch.equal(a[grp][name], a2[grp][name])
            e = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300)
            assert e <= 2e-4, (grp, name, e)


def _cluster_case(seed, R=11, K=20, O=10, B=400, D=8):
    rng = np.random.default_rng(seed)
    cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": "gaussian", "num_regions": R}
    params = {"params": {
        "rbf_list": {"centers": rng.uniform(-2, 2, size=(R, K, D)).astype(np.float32),
                     "log_sigs": rng.uniform(0.5, 1.5, size=(R, K)).astype(np.float32)},
        "linear": {"kernel": (rng.normal(size=(K, O)) * 0.3).astype(np.float32), "bias": (rng.normal(size=(O,)) * 0.1).astype(np.float32)},
        "cluster": {"kernel": rng.normal(size=(D, R)).astype(np.float32), "bias": rng.normal(size=(R,)).astype(np.float32)}}}
    x = rng.uniform(-2, 2, size=(B, D)).astype(np.float32)
    return rng, cfg, params, x


CLEAVES = LEAVES + (("cluster", "kernel"), ("cluster", "bias"))


def _t64(params, requires_grad=True):
    return {"params": {k: {n: torch.tensor(np.asarray(v, np.float64), requires_grad=requires_grad) for n, v in d.items()}
                       for k, d in params["params"].items()}}


@pytest.mark.parametrize("R,K,O,B", [(11, 20, 10, 400), (2, 64, 2, 70), (1, 33, 5, 130), (16, 8, 16, 1500)])
def test_cluster_wcrbfnet_vjp(gpu, R, K, O, B):
    """ClusterWCRBFNet (model.py:341-414): all six gradient leaves for cotangents of BOTH outputs (out, logits) against
    torch.autograd of the float64 restatement; without a logits cotangent; deterministic."""
    from irbfn_amd.model import ClusterWCRBFNet
    rng, cfg, params, x = _cluster_case(R * 3 + K, R, K, O, B)
    g = rng.normal(size=(B, O)).astype(np.float32)
    gl = rng.normal(size=(B, R)).astype(np.float32)
    net = ClusterWCRBFNet(**cfg)
    for with_logits in (True, False):
        a = net.vjp(params, torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda(),
                    glogits=torch.from_numpy(gl).cuda() if with_logits else None)["params"]
        a2 = net.vjp(params, torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda(),
                     glogits=torch.from_numpy(gl).cuda() if with_logits else None)["params"]
        tp = _t64(params)
        out, logits = orc.cluster_wcrbfnet_apply(cfg, tp, torch.tensor(x, dtype=torch.float64))
        obj = (out * torch.tensor(g, dtype=torch.float64)).sum()
        if with_logits:
            obj = obj + (logits * torch.tensor(gl, dtype=torch.float64)).sum()
        obj.backward()
        for grp, name in CLEAVES:
            ref, got = tp["params"][grp][name].grad.numpy(), a[grp][name].cpu().numpy()
            assert torch.equal(a[grp][name], a2[grp][name])
            e = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300)
            assert e <= 5e-5, (grp, name, with_logits, e)


def test_train_step_fullint_withcluster(gpu):
    """train_step_fullint_withcluster (scripts/train_nmpc_frenet.py:424-453): loss, the six-leaf gradient (through the
    softmax gate, the cross-entropy and integrate_frenet_mult) and the clip + Adam update against torch.autograd of
    the restatement; two steps."""
    from irbfn_amd.model import ClusterWCRBFNet
    R, T = 11, 5
    rng, cfg, params, x = _cluster_case(21, R=R, K=24, O=2 * T, B=300)
    B = x.shape[0]
    x[:, 7] = rng.normal(size=B).astype(np.float32) * 0.05
    x[:, 0] = rng.normal(size=B).astype(np.float32) * 0.2
    x[:, 2] = rng.uniform(1.0, 6.0, size=B).astype(np.float32)
    y = np.hstack([rng.normal(size=(B, T)) * 2, rng.normal(size=(B, T)) * 0.5]).astype(np.float32)
    ids = np.eye(R, dtype=np.float32)[rng.integers(0, R, size=B)]
    net = ClusterWCRBFNet(**cfg)
    state = train.ClusterTrainState.create(net, params, lr=1e-3, max_grad_norm=1.0)
    n = state.flat.numel()
    flat64 = lambda P: np.concatenate([np.asarray(P["params"][g_][n_], np.float64).reshape(-1) for g_, n_ in CLEAVES])
    assert n == flat64(params).size
    m, v, p_ref = np.zeros(n), np.zeros(n), flat64(params)
    cur = params
    for t in (1, 2):
        tp = _t64(cur)
        loss = orc.train_fullint_withcluster_loss(cfg, tp, torch.tensor(x, dtype=torch.float64), torch.tensor(y, dtype=torch.float64),
                                                  torch.tensor(ids, dtype=torch.float64), DP)
        loss.backward()
        g_ref = np.concatenate([tp["params"][g_][n_].grad.numpy().reshape(-1) for g_, n_ in CLEAVES])
        p_ref, m, v = orc.adam_update(p_ref, orc.clip_by_global_norm(g_ref, 1.0), m, v, t, lr=1e-3)
        state, l_gpu = train.train_step_fullint_withcluster(state, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(),
                                                            torch.from_numpy(ids).cuda(), DP)
        assert abs(float(l_gpu) - float(loss)) <= 3e-5 * abs(float(loss)), (t, float(l_gpu), float(loss))
        g_gpu = state.g.cpu().numpy()
        assert np.abs(g_gpu - g_ref).max() <= 5e-4 * np.abs(g_ref).max(), (t, np.abs(g_gpu - g_ref).max() / np.abs(g_ref).max())
        assert np.abs(state.flat.cpu().numpy() - p_ref).max() <= 5e-5 + 1e-6 * np.abs(p_ref).max()
        # next oracle step starts from the oracle's own parameters
        off, nxt = 0, {"params": {}}
        for g_, n_ in CLEAVES:
            shp = np.asarray(params["params"][g_][n_]).shape
            cnt = int(np.prod(shp))
            nxt["params"].setdefault(g_, {})[n_] = p_ref[off:off + cnt].reshape(shp)
            off += cnt
        cur = nxt


def test_train_step_two_ranks_unequal_shards(gpu, tmp_path):
    """The multi-rank branch of the train step (one all-reduce of [B_local * g, B_local * loss, B_local]): two ranks share
    cuda:0 over gloo with shards of 151 and 150 rows; parameters, gradient and loss after two steps equal the
    single-process steps on the whole 301-row batch (the global-batch mean, whatever the shard sizes)."""
    import os
    import subprocess
    import sys
    cfg = configs.model_card(3)
    P = configs.synth_params(3)
    B = 301
    x = configs.synth_queries(3, B=B)
    y = np.random.default_rng(8).normal(size=(B, cfg["out_features"])).astype(np.float32)
    p = P["params"]
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, x=x, y=y, centers=p["rbf_list"]["centers"], log_sigs=p["rbf_list"]["log_sigs"],
             kernel=p["linear"]["kernel"], bias=p["linear"]["bias"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29547", os.path.join(root, "tests", "_train_two_ranks.py"), inp, out],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    two = np.load(out)
    assert tuple(two["shard"]) == (0, 151)
    net = WCRBFNet.from_config(cfg)
    state = train.TrainState.create(net, P, lr=1e-3, max_grad_norm=1.0)
    losses = []
    for _ in range(2):
        state, loss = train.train_step_oneint(state, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), DP)
        losses.append(float(loss))
    g1, f1 = state.g.cpu().numpy(), state.flat.cpu().numpy()
    assert np.abs(two["losses"] - np.array(losses)).max() <= 2e-6 * np.abs(losses).max(), (two["losses"], losses)
    assert np.abs(two["g"] - g1).max() <= 2e-5 * np.abs(g1).max()
    assert np.abs(two["flat"] - f1).max() <= 2e-6 + 1e-6 * np.abs(f1).max()
