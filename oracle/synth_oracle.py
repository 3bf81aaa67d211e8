"""oracle/synth_oracle.py -- TEST INFRASTRUCTURE: host restatement of the DEVICE panel generator
(dynamic_factor_models_amd/csrc/synth.hip, dfm_synth_panels_dev), so that the panels the throughput runs are
generated from are checked cell by cell against the SURVEY.md §8(d) DGP and not by moments only.

The reference has no RNG and no Monte-Carlo loop (dfm_functions.ipynb has no `rand` call); there is nothing to cite
there.  What is restated: Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011, "Parallel random numbers: as easy as 1, 2,
3", the published algorithm and constants), keyed by (seed, replicate), one stream per quantity:
    stream 1 loadings (Box-Muller pairs), 2 idiosyncratic variances, 3 factor shocks, 4 idiosyncratic shocks,
    5 missing-cell uniforms
and the DGP on top of it:  lam_ij ~ N(0,1);  A = diag(linspace(.5,.9,r));  Q = I - AA';  R_i ~ U(.5,1.5);  f_0 ~ N(0,I);
f_t = A f_{t-1} + eta_t;  x_t = Lam f_t + sqrt(R) eps_t;  columns standardised as `standardize_data`
(dfm_functions.ipynb:501-509: mean and population s.d.) with the parameters rescaled to the standardised panel.
Only tests/ import this module.  Pinned by the Philox4x32-10 known-answer vectors of the Random123 distribution
(tests/test_oracle_synth.py).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
STR_LAM, STR_R, STR_F, STR_EPS, STR_MISS = 1, 2, 3, 4, 5


def philox4x32_10(counter, key):
    """counter: [..., 4] uint32, key: [2] uint32 (or [..., 2]) -> [..., 4] uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(np.asarray(key, dtype=np.uint32), c.shape[:-1] + (2,)), dtype=np.uint32, copy=True)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            n0 = hi1 ^ c[..., 1] ^ k[..., 0]
            n2 = hi0 ^ c[..., 3] ^ k[..., 1]
            c = np.stack([n0, lo1, n2, lo0], axis=-1)
            k = np.stack([k[..., 0] + W0, k[..., 1] + W1], axis=-1)
    return c


def _block(key64: int, stream: int, idx):
    """Philox::block(key, ctr_lo = idx, ctr_hi = stream) of dfm_philox.h for an array of indices."""
    idx = np.asarray(idx, dtype=np.uint64)
    ctr = np.stack([(idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                    np.full(idx.shape, stream & 0xFFFFFFFF, dtype=np.uint32),
                    np.full(idx.shape, (stream >> 32) & 0xFFFFFFFF, dtype=np.uint32)], axis=-1)
    key = np.array([key64 & 0xFFFFFFFF, (key64 >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key)


def _u01(a, b):
    x = (a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)
    return ((x >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal2(key64, stream, idx):
    o = _block(key64, stream, idx)
    u, v = _u01(o[..., 0], o[..., 1]), _u01(o[..., 2], o[..., 3])
    rad = np.sqrt(-2.0 * np.log(u))
    return rad * np.cos(2.0 * np.pi * v), rad * np.sin(2.0 * np.pi * v)


def uniform1(key64, stream, idx):
    o = _block(key64, stream, idx)
    return _u01(o[..., 0], o[..., 1])


def replicate_key(seed: int, replicate: int) -> int:
    return (seed ^ ((0x9E3779B97F4A7C15 * (replicate + 1)) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def synth_replicate_device(seed: int, replicate: int, N: int, T: int, r: int, missing: float = 0.0):
    """Replicate `replicate` of dfm_synth_panels_dev(seed, first_replicate, ...) (global index first_replicate + b).
    Returns (panel [T, N], dict(Lam, R, A, Q, mu0, P0))."""
    key = replicate_key(seed, replicate)
    n2 = (N * r + 1) // 2
    z0, z1 = normal2(key, STR_LAM, np.arange(n2))
    lam = np.empty(2 * n2)
    lam[0::2], lam[1::2] = z0, z1
    Lam = lam[:N * r].reshape(N, r).copy()
    Rv = 0.5 + uniform1(key, STR_R, np.arange(N))
    a = 0.5 + 0.4 * np.arange(r) / (r - 1) if r > 1 else np.array([0.5])
    q = np.sqrt(1.0 - a * a)
    f0, _ = normal2(key, STR_F, np.arange(r))
    F = np.empty((T, r))
    f = f0.copy()
    for t in range(0, T, 2):
        e0, e1 = normal2(key, STR_F, r + (t // 2) * r + np.arange(r))
        f = a * f + q * e0
        F[t] = f
        if t + 1 < T:
            f = a * f + q * e1
            F[t + 1] = f
    ncell = T * N
    e0, e1 = normal2(key, STR_EPS, np.arange((ncell + 1) // 2))
    eps = np.empty(2 * ((ncell + 1) // 2))
    eps[0::2], eps[1::2] = e0, e1
    eps = eps[:ncell].reshape(T, N)
    X = F @ Lam.T + np.sqrt(Rv) * eps
    m = X.sum(0) / T
    sd = np.sqrt(((X - m) ** 2).sum(0) / T)
    X = (X - m) / sd
    if missing > 0.0:
        u = uniform1(key, STR_MISS, np.arange(ncell)).reshape(T, N)
        X = np.where(u < missing, np.nan, X)
    A = np.diag(a)
    return X, dict(Lam=Lam / sd[:, None], R=Rv / sd ** 2, A=A, Q=np.diag(1.0 - a * a), mu0=np.zeros(r), P0=np.eye(r))
