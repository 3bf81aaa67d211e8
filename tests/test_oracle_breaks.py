"""Pins oracle/break_oracle.py (HAC / Chow / QLR, dfm_functions.ipynb:832-1047) to the reference notebook's saved
Table 4 (Stock_Watson.ipynb:1131-1157): rejection frequencies of the Chow and QLR tests and the quantiles of the
fitted-value correlations, 6 significant digits, r = 4 factors (r = 8 runs on the GPU, tests/test_gpu_breaks.py)."""
import json
import math
import os

import numpy as np

from oracle import break_oracle as bo

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "notebook_goldens.json")))["table4"]


def sig6(x, g):
    ulp = 10.0 ** (math.floor(math.log10(abs(g))) - 5)
    return abs(x - g) <= 0.5 * ulp * (1 + 1e-6)


def test_table4_r4():
    d = np.load(os.path.join(HERE, "golden", "sw_panel.npz"))
    o = bo.table4(d["bpdata"], d["inclcode"], 4)
    assert o["n"] == 176
    for lvl in range(3):
        assert sig6(o["chow_rej"][lvl], GOLD["chow_qlr_r4"][lvl][0]) and sig6(o["qlr_rej"][lvl], GOLD["chow_qlr_r4"][lvl][1])
    for x, g in zip(o["cor_pre"], GOLD["cor_r4"][0]):
        assert sig6(x, g)
    for x, g in zip(o["cor_post"], GOLD["cor_r4"][1]):
        assert sig6(x, g)


def test_hac_is_newey_west():
    """form_hscrc with the Bartlett kernel equals the textbook Newey-West sandwich."""
    g = np.random.default_rng(0)
    T, k, q = 90, 3, 4
    X = g.standard_normal((T, k)); u = g.standard_normal(T)
    z = X * u[:, None]
    S = z.T @ z
    for l in range(1, q + 1):
        G = z[l:].T @ z[:-l]
        S += (1 - l / (q + 1)) * (G + G.T)
    XXi = np.linalg.inv(X.T @ X)
    np.testing.assert_allclose(bo.form_hscrc(z, X, bo.form_kernel(q), q), XXi @ S @ XXi, rtol=1e-11)
