"""GPU parity of the batched Chow / QLR statistics (breaks.hip) against oracle/break_oracle.py and against the
reference notebook's Table 4 (Stock_Watson.ipynb:1131-1157) for r = 4 and r = 8 -- ALS on the three samples,
all (series x break date) regressions with HAC covariance, on the HIP kernels."""
import json
import math
import os

import numpy as np
import pytest

from oracle import break_oracle as bo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "notebook_goldens.json")))["table4"]


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def sig6(x, g):
    ulp = 10.0 ** (math.floor(math.log10(abs(g))) - 5)
    return abs(x - g) <= 0.5 * ulp * (1 + 1e-6)


@pytest.mark.parametrize("k,q", [(1, 0), (2, 3), (4, 6), (4, 0), (8, 6), (3, 15)])
def test_chow_batch_matches_oracle(ctx, k, q):
    g = np.random.default_rng(10 * k + q)
    ys, Xs, ps, pb, pq, want = [], [], [], [], [], []
    for s in range(5):
        T = 90 + 17 * s
        X = g.standard_normal((T, k)) + 0.3
        u = g.standard_normal(T)
        for t in range(1, T):
            u[t] += 0.5 * u[t - 1]                                   # serially correlated errors
        y = X @ g.standard_normal(k) + u
        y[T // 2:] += X[T // 2:, 0]                                  # a break
        ys.append(y); Xs.append(X)
        for tb in (T // 4, T // 2, T - T // 5):
            ps.append(s); pb.append(tb); pq.append(q)
            want.append(bo.compute_chow(y, X, q, tb))
    got = ctx.chow_batch_host(ys, Xs, ps, pb, pq)
    np.testing.assert_allclose(got, want, rtol=1e-8)


@pytest.mark.parametrize("r", [4, 8])
def test_table4_on_the_gpu(ctx, r):
    from dynamic_factor_models_amd import api
    d = np.load(os.path.join(HERE, "golden", "sw_panel.npz"))
    bp, inc = d["bpdata"], d["inclcode"]
    ms = [api.DFMModel(bp, inc, 20, 40, a, b, 0, r, 1e-8, 4, 4) for a, b in ((3, 224), (3, 104), (105, 224))]
    for m in ms:
        api.estimate_factor(m, computeR2=False, ctx=ctx)
    chow, qlr = api.break_tests(ms[0], 104, ctx=ctx)
    o = bo.table4(bp, inc, r, factors=tuple(m.factor for m in ms), stats=(chow, qlr))
    key = f"chow_qlr_r{r}"
    assert o["n"] == 176
    for lvl in range(3):
        assert sig6(o["chow_rej"][lvl], GOLD[key][lvl][0]), (lvl, o["chow_rej"], GOLD[key])
        assert sig6(o["qlr_rej"][lvl], GOLD[key][lvl][1]), (lvl, o["qlr_rej"], GOLD[key])
    for x, g in zip(o["cor_pre"], GOLD[f"cor_r{r}"][0]):
        assert sig6(x, g), (o["cor_pre"], GOLD[f"cor_r{r}"])
    for x, g in zip(o["cor_post"], GOLD[f"cor_r{r}"][1]):
        assert sig6(x, g)
    if r == 4:                                                       # statistic by statistic against the oracle
        ref = bo.table4(bp, inc, 4)
        np.testing.assert_allclose(chow, ref["chow"], rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(qlr, ref["qlr"], rtol=1e-6, equal_nan=True)
