"""CPU restatement of ``Sampler._compute_evidence`` (``pocomc/sampler.py:869-920``) -- TEST INFRASTRUCTURE ONLY (imported by
``tests/``; the product never touches ``oracle/``).

Importance sampling of the evidence with the trained flow as the proposal:

    theta_q, logq = flow.sample(n)                 :886-889
    x_q, logdetj  = scaler.inverse(theta_q)        :892
    logp          = log_prior(x_q)                 :895, non-finite rows dropped :898-901
    logl          = log_like(x_q)                  :904
    logw          = logl + logp + logdetj - logq   :907
    logz          = logaddexp.reduce(logw) - log(len(logw))                                   :910
    dlogz         = std of max(n, 1000) bootstrap replicates of that estimate                 :913

The reference draws the base sample from torch's stream and the bootstrap indices from numpy's legacy stream
(``np.random.choice(m, m)`` per replicate); both are inputs here (``z``) or recorded (``draws``) so that the device path
can replay them.  The flow is ``oracle.maf.OracleMAF`` (parity unpinned, see its header); everything else follows the
reference line by line and is pinned through ``oracle.scaler`` (golden vectors from the reference itself).
"""
import numpy as np


def compute_evidence(maf, scaler, log_prior, log_like, z, seed=0, n_boot=None):
    """``z``: (n, D) float32 base draw.  Returns ``(logz, dlogz, draws, logw)``; ``draws``: (B, m) int64, the bootstrap
    indices as ``np.random.seed(seed)`` + the reference's calls produce them."""
    n = len(z)
    theta_q, logq = maf.sample_from(np.asarray(z, dtype=np.float32))          # :886-889 (float32 flow)
    theta_q = theta_q.astype(np.float64)                                       # torch_to_numpy: float64 arrays downstream
    logq = logq.astype(np.float64)
    x_q, logdetj = scaler.inverse(theta_q)                                     # :892
    logp = log_prior(x_q)                                                      # :895
    ok = np.isfinite(logp)                                                     # :898-901
    x_q, logdetj, logq, logp = x_q[ok], logdetj[ok], logq[ok], logp[ok]
    logl = log_like(x_q)                                                       # :904
    logw = logl + logp + logdetj - logq                                        # :907
    m = len(logw)
    logz = np.logaddexp.reduce(logw) - np.log(m)                               # :910
    B = int(np.maximum(n, 1000)) if n_boot is None else int(n_boot)            # :913
    np.random.seed(seed)
    draws = np.stack([np.random.choice(m, m) for _ in range(B)])
    dlogz = np.std([np.logaddexp.reduce(logw[d]) - np.log(m) for d in draws])
    return float(logz), float(dlogz), draws.astype(np.int64), logw
