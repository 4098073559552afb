"""Generates the committed golden fixtures under tests/golden/ (run once in the build container,
where /root/reference is mounted; the GPU box only sees the generated files).

* ``kat.json``        KAT-1 (scripts/test_dynamics.ipynb cells 1-4, recorded float32 output),
                      KAT-2 / KAT-3 (CommonRoad derivative vectors,
                      deprecated/f1tenth_gym/tests/test_dynamics.py:32-39,55-96) -- DATA copied from
                      the reference's own test/notebook, no source text.
* ``ckpt_<run>.npz``  trained parameters decoded from the reference's Flax msgpack checkpoints
                      (scripts/ckpts/<run>/checkpoint_<n>; ext type 1 = ndarray) with plain ``msgpack``
                      (nothing is unpickled / executed), the YAML model card next to it as JSON, 64
                      queries drawn U(lower, upper) with seed 123, and the float64 oracle outputs.
* ``synth_cfg1.npz``  BASELINE config 1 (256 centres, d=3, B=1024) float64 oracle output.

The reference itself cannot be imported here (JAX / Flax are not installed: ordinary
ModuleNotFoundError), so expected outputs come from the CPU restatement in oracle/ --
"parity unpinned" for everything except the roll-out KATs (see oracle/irbfn_oracle.py header).
"""
import ast
import json
import os
import sys

import msgpack
import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import irbfn_oracle as orc  # noqa: E402
from irbfn_amd import configs  # noqa: E402

REF = "/root/reference"
CKPTS = [
    ("dnmpc_1regions_newdata_oldintloss_nomirror_highk", 999),
    ("dnmpc_128regions", 600),
    ("dnmpc_1regions_newnewdata_1stepst_l1_newarch_ksint_iq", 5300),
    ("dnmpc_12regions_frenet_l1_bigdata", 9500),
]
CFG_KEYS = ("in_features", "out_features", "num_kernels", "basis_func", "num_regions", "lower_bounds",
            "upper_bounds", "dimension_ranges", "activation_idx", "delta", "mu", "cs", "seed")


def _ext_hook(code, data):
    if code == 1:   # flax.serialization ndarray: (shape, dtype name, raw bytes)
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
    return msgpack.ExtType(code, data)


def load_flax_msgpack(path):
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)


def kat():
    nb = json.load(open(os.path.join(REF, "scripts/test_dynamics.ipynb")))
    cell4 = [c for c in nb["cells"] if "".join(c["source"]).strip() == "all_states"][0]
    txt = "".join(cell4["outputs"][0]["data"]["text/plain"])
    txt = txt[txt.index("["):txt.rindex("]") + 1]
    arr = np.array(ast.literal_eval(txt.replace("\n", "")), dtype=np.float32)   # nested float lists
    assert arr.shape == (10, 5, 7)
    return {
        "kat1": {"source": "scripts/test_dynamics.ipynb cells 1-4 (float32, recorded on a CUDA GPU)",
                 "dyn_params": orc.DYN_PARAMS_KAT1, "states0": "zeros(10,7)", "u": -10.0,
                 "all_states_row": arr[0].astype(np.float64).tolist(),
                 "rows_identical": bool((arr == arr[0]).all())},
        "kat2": {"source": "deprecated/f1tenth_gym/tests/test_dynamics.py:32-39,62-70,83-96",
                 "vehicle": {"mu": 1.0489, "C_Sf": 21.92 / 1.0489, "C_Sr": 21.92 / 1.0489,
                             "lf": 0.3048 * 3.793293, "lr": 0.3048 * 4.667707, "h": 0.3048 * 2.01355,
                             "m": 4.4482216152605 / 0.3048 * 74.91452, "I": 4.4482216152605 * 0.3048 * 1321.416},
                 "x_st": [2.0233348142065677, 0.0041907137716636, 0.0197545248559617, 15.7216236334290116,
                          0.0025857914776859, 0.0529001056654038, 0.0033012170610298],
                 "u": [0.15, 5.3536773276413925],
                 "f_st_gt": [15.7213512030862397, 0.0925527979719355, 0.1500000000000000, 5.3536773276413925,
                             0.0529001056654038, 0.6435589397748606, 0.0313297971641291]},
        "kat3": {"source": "deprecated/f1tenth_gym/tests/test_dynamics.py:55-61,74-82",
                 "x_ks": [3.9579422297936526, 0.0391650102771405, 0.0378491427211811, 16.3546957860883566,
                          0.0294717351052816],
                 "u": [0.15, 5.1464424102339752],
                 "f_ks_gt": [16.3475935934250209, 0.4819314886013121, 0.1500000000000000, 5.1464424102339752,
                             0.2401426578627629]},
    }


def ckpt_fixture(run, step):
    cfg = yaml.safe_load(open(os.path.join(REF, "scripts/configs", run + ".yaml")))
    cfg = {k: cfg[k] for k in CFG_KEYS if k in cfg}
    tree = load_flax_msgpack(os.path.join(REF, "scripts/ckpts", run, f"checkpoint_{step}"))
    p = tree["params"]["params"]
    params = {"params": {"rbf_list": {"centers": p["rbf_list"]["centers"], "log_sigs": p["rbf_list"]["log_sigs"]},
                         "linear": {"kernel": p["linear"]["kernel"], "bias": p["linear"]["bias"]}}}
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)])
    hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    rng = np.random.default_rng(123)
    x = rng.uniform(lo, hi, size=(64, cfg["in_features"]))
    p64 = orc.cast_params(params, np.float64)
    out, h, gam = orc.wcrbfnet_apply(cfg, p64, x, return_aux=True)
    np.savez_compressed(
        os.path.join(HERE, f"ckpt_{run}.npz"),
        centers=p["rbf_list"]["centers"], log_sigs=p["rbf_list"]["log_sigs"], kernel=p["linear"]["kernel"],
        bias=p["linear"]["bias"], x=x, out64=out, h64=h, gamma64=gam, step=np.int64(step))
    with open(os.path.join(HERE, f"ckpt_{run}.json"), "w") as f:
        json.dump(cfg, f)
    print(run, "out range", float(out.min()), float(out.max()), "gamma>1e-3 per query",
          float((gam > 1e-3).sum(1).mean()))


def deeper_fixture(run="dnmpc_1regions_frenet_l1_bigdata_5stepint_deeper", step=9999):
    """DeeperWCRBFNet checkpoint (the model of IRBFNFrenetPlanner(deeper=True)) + float64 oracle outputs."""
    cfg = yaml.safe_load(open(os.path.join(REF, "scripts/configs", run + ".yaml")))
    cfg = {k: cfg[k] for k in CFG_KEYS if k in cfg}
    tree = load_flax_msgpack(os.path.join(REF, "scripts/ckpts", run, f"checkpoint_{step}"))
    p = tree["params"]["params"]
    params = {"params": {k: {n: np.asarray(v, np.float64) for n, v in p[k].items()}
                         for k in ("rbf_list", "linear_pre1", "linear_pre2", "linear")}}
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)])
    hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = np.random.default_rng(123).uniform(lo, hi, size=(64, cfg["in_features"]))
    out = orc.deeper_wcrbfnet_apply(cfg, params, x)
    flat = {f"{k}__{n}": np.asarray(p[k][n]) for k in ("rbf_list", "linear_pre1", "linear_pre2", "linear") for n in p[k]}
    np.savez_compressed(os.path.join(HERE, f"ckpt_{run}.npz"), x=x, out64=out, file_step=np.int64(step),
                        step=np.int64(tree["step"]), **flat)
    with open(os.path.join(HERE, f"ckpt_{run}.json"), "w") as f:
        json.dump(cfg, f)
    print(run, "out range", float(out.min()), float(out.max()))


def synth_cfg1():
    cfg = configs.model_card(1)
    params = configs.synth_params(1, np.float64)
    x = configs.synth_queries(1, dtype=np.float64)
    out = orc.wcrbfnet_apply(cfg, params, x)
    np.savez_compressed(os.path.join(HERE, "synth_cfg1.npz"), out64=out)
    print("cfg1 out range", float(out.min()), float(out.max()))


if __name__ == "__main__":
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat(), f, indent=1)
    for run, step in CKPTS:
        ckpt_fixture(run, step)
    synth_cfg1()
    deeper_fixture()
