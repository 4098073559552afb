"""Basis-function tokens with the names of the reference's ``flax_rbf`` kernels
(deprecated/f1tenth_gym/examples/flax_rbf/flax_rbf/flax_rbf.py:34-111).

The reference passes ``basis_func=gaussian`` (a callable, restored by ``eval(conf.basis_func)`` at
src/irbfn_mpc/irbfn_planner.py:72).  Here the tokens only *name* the kernel; the arithmetic runs in
the HIP kernels.  ``WCRBFNet(basis_func=...)`` accepts a token or its string name (no ``eval``).
"""
from __future__ import annotations


class BasisFunc:
    __slots__ = ("name",)

    def __init__(self, name: str):
        self.name = name

    def __repr__(self):
        return self.name

    def __call__(self, alpha):
        raise RuntimeError(
            f"{self.name} is a token naming a HIP kernel, not a host function; evaluate it through "
            "WCRBFNet.apply (the CPU restatement lives in oracle/, test-only)")


NAMES = ("gaussian", "gaussian_wide", "gaussian_wider", "inverse_quadratic", "linear", "quadratic",
         "multiquadric", "inverse_multiquadric", "spline", "poisson_one", "poisson_two", "matern32",
         "matern52")
for _n in NAMES:
    globals()[_n] = BasisFunc(_n)
__all__ = list(NAMES) + ["BasisFunc", "basis_name"]


def basis_name(b) -> str:
    name = b.name if isinstance(b, BasisFunc) else getattr(b, "__name__", b)
    if not isinstance(name, str) or name not in NAMES:
        raise ValueError(f"unknown basis function {b!r}; known: {NAMES} "
                         "(gaussian_narrow[er] exist only upstream, no source in the reference snapshot)")
    return name
