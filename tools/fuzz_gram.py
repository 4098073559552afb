"""Random nets through K1g (forward, against the float64 restatement in numpy below -- tools may not import oracle/) and K2g (parameter
VJP, against the all-float32 K2): worst errors, non-finite outputs, run-to-run differences.  python tools/fuzz_gram.py [cases] [seed] [wide]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402

BASES = {"gaussian": lambda t: np.exp(-t), "gaussian_wide": lambda t: np.exp(-0.1 * t), "inverse_quadratic": lambda t: 1.0 / (1.0 + t),
         "inverse_multiquadric": lambda t: 1.0 / np.sqrt(1.0 + t)}


def ref_forward(cfg, P, x):
    c = P["params"]["rbf_list"]["centers"][0].astype(np.float64); ls = P["params"]["rbf_list"]["log_sigs"][0].astype(np.float64)
    W = P["params"]["linear"]["kernel"].astype(np.float64); b = P["params"]["linear"]["bias"].astype(np.float64)
    x = x.astype(np.float64)
    d2 = ((x[:, None, :] - c[None, :, :]) ** 2).sum(-1) * np.exp(-2.0 * ls)[None, :]
    phi = BASES[cfg["basis_func"]](d2)
    lo = np.array([v[0] for v in cfg["lower_bounds"]]); hi = np.array([v[0] for v in cfg["upper_bounds"]]); dl = np.array(cfg["delta"])
    gam = np.prod((np.tanh(dl * (x - lo)) + 1) / 2 * (np.tanh(dl * (hi - x)) + 1) / 2, axis=1)
    return (gam[:, None] * phi) @ W + b, (gam[:, None] * phi) @ np.abs(W) + np.abs(b)


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
    worst_f, worst_v, bad, names = 0.0, 0.0, [], {}
    for case in range(ncase):
        D = int(rng.choice([2, 3, 4, 5, 6, 7, 8])); K = int(rng.choice([17, 32, 50, 100, 129, 200, 333, 512, 1000])); O = int(rng.choice([1, 2, 5, 10, 11, 16]))
        if wide:                                   # K1g's wide form (16 < O <= 128, d = 7 / 8, >= 256 centres, >= 2048 queries; forward only)
            D = int(rng.choice([7, 8])); K = int(rng.choice([256, 300, 512, 1000])); O = int(rng.choice([17, 20, 33, 64, 100, 128]))
        basis = str(rng.choice(list(BASES)))
        span = float(rng.choice([0.5, 2.0, 10.0])); off = float(rng.choice([0.0, 3.0, -20.0]))
        lo, hi = np.full(D, off - span), np.full(D, off + span)
        cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": basis, "num_regions": 1,
               "lower_bounds": [[float(v)] for v in lo], "upper_bounds": [[float(v)] for v in hi], "dimension_ranges": [[0] * D],
               "activation_idx": list(range(D)), "delta": [float(rng.choice([5.0, 20.0, 100.0]))] * D}
        sig = float(rng.choice([0.3, 1.0, 3.0])) * span
        P = {"params": {"rbf_list": {"centers": rng.uniform(lo - 0.3 * span, hi + 0.3 * span, size=(1, K, D)).astype(np.float32),
                                     "log_sigs": (np.log(sig) + rng.uniform(-0.7, 0.7, size=(1, K))).astype(np.float32)},
                        "linear": {"kernel": (rng.normal(size=(K, O)) * 10.0 ** rng.uniform(-3, 2, size=(1, O))).astype(np.float32),
                                   "bias": rng.normal(size=(O,)).astype(np.float32)}}}
        net = WCRBFNet.from_config(cfg)
        B = int(rng.choice([2048, 2500, 4100])) if wide else int(rng.choice([70, 1500, 2100]))
        x = rng.uniform(lo - 0.1 * span, hi + 0.1 * span, size=(B, D)).astype(np.float32)
        tag = f"case {case}: D={D} K={K} O={O} {basis} span={span} off={off} sig={sig:.2f} B={B}"
        try:
            net.set_options(fwd_kernel=_lib.FWD_K1G)
            got = net.apply(P, x); name = net.last_launch()["kernel"]
            got2 = net.apply(P, x)
        except ValueError as e:                    # parameters outside the expansion's budget: refused, as designed
            net.set_options(fwd_kernel=_lib.FWD_AUTO)
            continue
        net.set_options(fwd_kernel=_lib.FWD_K1)
        k1 = net.apply(P, x)                       # the all-float32 kernel on the same inputs: the gate's own float32 sensitivity shows there too
        net.set_options(fwd_kernel=_lib.FWD_AUTO)
        ref, scale = ref_forward(cfg, P, x)
        err = float((np.abs(got - ref) / (scale + 1e-300)).max())
        err1 = float((np.abs(k1 - ref) / (scale + 1e-300)).max())
        if not np.isfinite(got).all() or not np.array_equal(got, got2) or err > max(3e-6, 3.0 * err1):
            bad.append((tag, "forward", name, f"K1g {err:.2e}", f"K1 {err1:.2e}"))
        worst_f = max(worst_f, err)
        names[name.split("<")[0]] = names.get(name.split("<")[0], 0) + 1
        if wide:
            continue
        xt = torch.from_numpy(x).cuda(); g = torch.from_numpy(rng.normal(size=(B, O)).astype(np.float32)).cuda()
        try:
            net.set_options(vjp_kernel=_lib.VJP_K2G)
            a = net.vjp(P, xt, g)["params"]; a2 = net.vjp(P, xt, g)["params"]
        except ValueError:
            net.set_options(vjp_kernel=_lib.VJP_AUTO)
            continue
        net.set_options(vjp_kernel=_lib.VJP_K2)
        r = net.vjp(P, xt, g)["params"]
        net.set_options(vjp_kernel=_lib.VJP_AUTO)
        for grp, nm in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
            e = float((a[grp][nm] - r[grp][nm]).abs().max() / (r[grp][nm].abs().max() + 1e-30))
            if not torch.isfinite(a[grp][nm]).all() or not torch.equal(a[grp][nm], a2[grp][nm]) or e > 2e-5:
                bad.append((tag, "vjp " + nm, e))
            worst_v = max(worst_v, e)
    print("forward kernels:", names)
    print(f"{ncase} cases: worst forward error / sum|terms| {worst_f:.2e}; worst VJP error / max|leaf| vs K2 {worst_v:.2e}; flagged: {len(bad)}")
    for b in bad[:20]:
        print("  ", b)


if __name__ == "__main__":
    main()
