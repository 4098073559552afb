"""irbfn_amd.jax_adapter: import-guarded.  JAX is not installed in the build image, so the numerical test
below runs only where a maintainer of the reference has both JAX and a GPU; here the module must import and fail loudly."""
import pytest


def test_adapter_imports_and_reports_missing_jax():
    from irbfn_amd import configs, jax_adapter
    from irbfn_amd.model import WCRBFNet
    net = WCRBFNet.from_config(configs.model_card(1))
    try:
        import jax  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="needs jax"):
            jax_adapter.make_irbfn(net)
        return
    assert callable(jax_adapter.make_irbfn(net))


@pytest.mark.gpu
def test_grad_through_the_adapter_matches_the_vjp(gpu):
    jax = pytest.importorskip("jax")
    import numpy as np
    from irbfn_amd import configs, jax_adapter
    from irbfn_amd.model import WCRBFNet
    cfg, P = configs.model_card(1), configs.synth_params(1)
    net = WCRBFNet.from_config(cfg)
    irbfn = jax_adapter.make_irbfn(net)
    x = configs.synth_queries(1, B=64)
    g = configs.synth_cotangent(1, B=64)
    out = irbfn(P, x)
    assert np.allclose(np.asarray(out), net.apply(P, x), rtol=1e-6)
    grads = jax.grad(lambda p: (irbfn(p, x) * g).sum())(P)
    ref = net.vjp(P, x, g)
    assert np.allclose(np.asarray(grads["params"]["linear"]["kernel"]), ref["params"]["linear"]["kernel"], rtol=1e-5, atol=1e-6)
