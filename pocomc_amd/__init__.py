"""pocomc_amd -- MI355X-native engine for pocoMC's flow-preconditioned MCMC step.

``Flow`` and the four MCMC kernels keep the reference's Python surface
(``pocomc/flow.py``, ``pocomc/mcmc.py``); their arithmetic runs in hand-written
gfx950 kernels behind the C ABI of ``include/pocomc_amd.h``.
"""
from .maf_spec import MAFSpec  # noqa: F401

__all__ = ["Flow", "MAFSpec", "Reparameterize", "Sampler", "Prior", "mcmc", "tools"]


def __getattr__(name):
    # torch / the shared library are only touched when the GPU classes are used
    if name == "Flow":
        from .flow import Flow
        return Flow
    if name == "Reparameterize":
        from .scaler import Reparameterize
        return Reparameterize
    if name in ("mcmc", "tools", "geometry", "prior", "sampler", "train"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name == "Sampler":
        from .sampler import Sampler
        return Sampler
    if name == "Prior":
        from .prior import Prior
        return Prior
    raise AttributeError(name)
