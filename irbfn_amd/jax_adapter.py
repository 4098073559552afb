"""``irbfn(params, x)`` / ``jax.grad`` of it, for code that stays in JAX (the reference's call surface:
``state.apply_fn(state.params, x)`` inside jitted train / predict steps, src/irbfn_mpc/irbfn_planner.py:29-32,
scripts/train_nmpc.py:268-299).

The forward and the parameter VJP run in the HIP kernels behind ``WCRBFNet``; JAX sees one primitive with a custom VJP that
calls out through ``jax.pure_callback``.  JAX is NOT part of the build image (nothing here can be exercised there:
``tests/test_jax_adapter_cpu.py`` skips without it), so this module is the written-out form of INTEGRATION.md section 3 and
imports JAX only when used."""
from __future__ import annotations

import numpy as np


def make_irbfn(net):
    """net: ``irbfn_amd.model.WCRBFNet``.  Returns ``irbfn(params, x) -> out[B, O]`` usable under ``jax.jit`` and ``jax.grad``
    (gradients w.r.t. ``params``; the reference never differentiates w.r.t. ``x`` -- zeros are returned for it)."""
    try:
        import jax
        import jax.numpy as jnp
    except ImportError as e:                    # pragma: no cover - JAX is absent in the build image
        raise ImportError("irbfn_amd.jax_adapter needs jax (the reference's own dependency); it is not installed here") from e
    dtype = jnp.float64 if net.use_float64 else jnp.float32
    host = lambda tree: jax.tree_util.tree_map(np.asarray, tree)

    @jax.custom_vjp
    def irbfn(params, x):
        shape = jax.ShapeDtypeStruct((x.shape[0], net.out_features), dtype)
        return jax.pure_callback(lambda p, xx: np.asarray(net.apply(host(p), np.asarray(xx)), dtype), shape, params, x)

    def fwd(params, x):
        return irbfn(params, x), (params, x)

    def bwd(res, g):
        params, x = res
        shapes = jax.tree_util.tree_map(lambda a: jax.ShapeDtypeStruct(a.shape, dtype), params)

        def cb(p, xx, gg):
            grads = net.vjp(host(p), np.asarray(xx), np.asarray(gg))
            inner = grads["params"]
            # same nesting as the caller's pytree: {"params": {...}} or the inner dict
            out = grads if "params" in p else inner
            return jax.tree_util.tree_map(lambda a: np.asarray(a, dtype), out)
        grads = jax.pure_callback(cb, shapes, params, x, g)
        return grads, jnp.zeros_like(x)

    irbfn.defvjp(fwd, bwd)
    return irbfn
