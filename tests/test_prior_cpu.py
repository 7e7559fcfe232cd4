"""``pocomc_amd.Prior`` against the reference's own tests (``tests/test_prior.py``: shapes, finiteness, bounds, dim) and
against scipy, on the CPU -- the class is host-side; its device evaluation is checked in ``test_gpu_tools.py``."""
import numpy as np
from scipy.stats import norm, uniform

from pocomc_amd.prior import Prior


def test_sample_shape():                                   # tests/test_prior.py:10-13
    x = Prior([norm(0, 1), norm(0, 1)]).rvs(10)
    assert np.shape(x) == (10, 2)


def test_log_prob_array_shape_sign_finite():               # tests/test_prior.py:15-37
    prior = Prior([norm(0, 1), norm(0, 1)])
    lp = prior.logpdf(prior.rvs(10))
    assert isinstance(lp, np.ndarray) and np.shape(lp) == (10,)
    assert np.all(lp < 0) and np.all(np.isfinite(lp))


def test_bounds_and_dim():                                 # tests/test_prior.py:39-51
    prior = Prior([norm(0, 1), norm(0, 1)])
    b = prior.bounds
    assert np.shape(b) == (2, 2) and np.all(b[:, 0] < b[:, 1])
    assert prior.dim == 2
    b = Prior([uniform(-3, 5), norm(1, 2)]).bounds
    assert np.array_equal(b[0], [-3.0, 2.0]) and np.isinf(b[1]).all()


def test_logpdf_is_the_sum_of_the_factors_and_minus_inf_outside():   # prior.py:70-100
    dists = [uniform(-3, 5), norm(1, 2), uniform(0, 1)]
    prior = Prior(dists)
    rng = np.random.default_rng(0)
    x = np.column_stack([rng.uniform(-4, 3, 50), rng.normal(1, 2, 50), rng.uniform(-0.5, 1.5, 50)])
    want = sum(d.logpdf(x[:, j]) for j, d in enumerate(dists))
    got = prior.logpdf(x)
    inside = np.isfinite(want)
    np.testing.assert_allclose(got[inside], want[inside], rtol=1e-13)
    assert (got[~inside] == -np.inf).all() and (~inside).any() and inside.any()
