"""Flow training loop -- host-side mirror of ``pocomc/flow.py:165-384`` (placeholder:
the fwd+bwd+AdamW kernels are the next milestone)."""


def fit_flow(flow, x, weights=None, **kwargs):
    raise NotImplementedError("Flow.fit: the gfx950 training kernels are not built yet")
