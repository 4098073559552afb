"""Builds libirbfn_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

``python -m irbfn_amd.build`` or ``irbfn_amd.build.build_lib()``.  hipcc cross-compiles without a
GPU.  Objects are cached under ``irbfn_amd/csrc/_obj`` and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libirbfn_hip.so")
ARCH = "gfx950"

_SLP = ["-fno-slp-vectorize"] if os.environ.get("IRBFN_NO_SLP") == "1" else []

# (source, object name, extra flags)
UNITS = [
    ("abi.hip", "abi.o", []),
    ("pack_all.hip", "pack_all.o", []),          # K0: every image of a net in two launches
    ("rbf_forward.hip", "rbf_forward.o", []),
    # NOTE on -fno-slp-vectorize (IRBFN_NO_SLP=1): hipcc's SLP vectoriser fuses the weight-row FMAs
    # into v_pk_fma_f32 with an SGPR-pair operand; measured on MI355X (cfg-2) the packed form is
    # FASTER than ten v_fmac_f32 with SGPR operands (152 vs 173 us), so SLP stays on by default.
    ("rbf_forward_kernels.hip", "rbf_fwd_d3.o", ["-DIRBFN_INST_D=3"] + _SLP),
    ("rbf_forward_kernels.hip", "rbf_fwd_d4.o", ["-DIRBFN_INST_D=4"] + _SLP),
    ("rbf_forward_kernels.hip", "rbf_fwd_d7.o", ["-DIRBFN_INST_D=7"] + _SLP),
    ("rbf_forward_kernels.hip", "rbf_fwd_d8.o", ["-DIRBFN_INST_D=8"] + _SLP),
    ("rbf_forward_mfma.hip", "rbf_fwd_mfma.o", []),
    ("rbf_forward_f16.hip", "rbf_fwd_f16.o", []),
    ("rbf_forward_small.hip", "rbf_fwd_small.o", []),
    # region-sparse kernels of the multi-region nets (per-lane lists of active regions)
    ("rbf_sparse.hip", "rbf_sparse.o", ["-fno-slp-vectorize"]),
    # float64 mode (--use_float64 of the reference): plain f64 VALU kernels
    ("rbf_f64.hip", "rbf_f64.o", []),
    ("rbf_forward_gram.hip", "rbf_forward_gram.o", []),
    ("rbf_forward_gram_wide.hip", "rbf_forward_gram_wide.o", []),
    ("rbf_vjp.hip", "rbf_vjp.o", [] + _SLP),
    ("rbf_vjp_f16.hip", "rbf_vjp_f16.o", ["-fno-slp-vectorize"]),
    ("rbf_vjp_gram.hip", "rbf_vjp_gram.o", ["-fno-slp-vectorize"]),   # VGPR operands: plain FMAs (2.4 cyc) beat packed (4.7) + pairing moves
    # 12-step unrolled groups of the roll-out; no SLP: v_pk_* cost more than the two plain VALU instructions they replace
    ("rollout.hip", "rollout.o", ["-mllvm", "-pragma-unroll-threshold=100000", "-fno-slp-vectorize"]),
    # the fused planning tick: wide K1h forward + K3p roll-out core in one kernel (the roll-out's flags: 50-knot register arrays)
    ("plan_tick_wide.hip", "plan_tick_wide.o", ["-mllvm", "-pragma-unroll-threshold=100000", "-fno-slp-vectorize"]),
    ("rollout_vjp.hip", "rollout_vjp.o", ["-mllvm", "-pragma-unroll-threshold=100000", "-fno-slp-vectorize"]),
    ("train_step.hip", "train_step.o", []),
    ("mlp_head.hip", "mlp_head.o", []),
    ("planner_front.hip", "planner_front.o", []),
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(INCLUDE, "irbfn_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(unit, hipcc, hdr_mtime, force):
    src, obj, extra = unit
    srcp, objp = os.path.join(CSRC, src), os.path.join(OBJ, obj)
    if (not force and os.path.exists(objp)
            and os.path.getmtime(objp) >= max(os.path.getmtime(srcp), hdr_mtime)):
        return objp, False
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-I", INCLUDE, "-I", CSRC,
           *extra, "-c", srcp, "-o", objp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src} {extra}:\n{r.stderr[-4000:]}")
    return objp, True


def build_lib(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdr = _newest_header()
    jobs = jobs or min(len(UNITS), os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda u: _compile(u, hipcc, hdr, force), UNITS))
    objs = [r[0] for r in res]
    rebuilt = any(r[1] for r in res)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[irbfn_amd.build] {'rebuilt' if rebuilt else 'up to date'}: {LIB}")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, verbose=True)
