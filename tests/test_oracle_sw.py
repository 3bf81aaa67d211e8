"""Pins the oracle to the reference's own known-answer values.

The reference has no tests; the numbers its author saved in `Stock_Watson.ipynb`'s cell outputs are the only
golden vectors that exist.  `tests/golden/notebook_goldens.json` is their machine transcription
(`tests/golden/make_notebook_goldens.py`, raw-line citations inside) and `tests/golden/sw_panel.npz` is the
panel produced by the restated ingestion (`oracle/sw_panel.py` <- `readin_functions.jl`).  Every printed
digit of Tables 2(A), 2(B), 2(C), 3 and 5 must be reproduced by oracle/als_oracle.py.
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import als_oracle as ao

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "notebook_goldens.json")))
REF_XLSX = "/root/reference/data/hom_fac_1.xlsx"
INIT, LAST = 3, 224          # (1959,3)..(2014,4): Stock_Watson.ipynb:217-218


@pytest.fixture(scope="module")
def sw():
    d = np.load(os.path.join(HERE, "golden", "sw_panel.npz"))
    bp, inc, cat = d["bpdata"], d["inclcode"], d["bpcatcode"]
    real = np.isin(np.floor(cat), [1, 2, 3, 5])                 # readin_functions.jl:254
    return dict(all=bp, inc_all=inc, real=bp[:, real], inc_real=inc[real], names=[str(s) for s in d["bpnamevec"]])


def _shown(x, digits=3):
    """True value x printed by `round(x, digits=3)`."""
    return round(float(x), digits)


def _sig6(x, g):
    """x agrees with g, a value Julia printed with 6 significant digits."""
    if g == 0:
        return abs(x) < 5e-7
    ulp = 10.0 ** (math.floor(math.log10(abs(g))) - 5)
    return abs(x - g) <= 0.5 * ulp * (1 + 1e-6)


def test_fixture_dimensions(sw):
    """panel sizes, include counts, missing cells, in-window balance."""
    assert sw["all"].shape == (224, 207) and sw["real"].shape == (224, 86)
    assert (sw["inc_all"] == 1).sum() == 139 and (sw["inc_real"] == 1).sum() == 58
    assert np.isnan(sw["all"]).sum() == 2374 and np.isnan(sw["real"]).sum() == 323
    xa = sw["all"][INIT - 1: LAST][:, sw["inc_all"] == 1]
    xr = sw["real"][INIT - 1: LAST][:, sw["inc_real"] == 1]
    assert (~np.isnan(xa)).sum() == 29098 and (~np.isnan(xa)).all(0).sum() == 94
    assert (~np.isnan(xr)).sum() == 12700 and (~np.isnan(xr)).all(0).sum() == 49
    assert GOLD["dims"]["quarterly"] == [224, 85]


@pytest.mark.skipif(not os.path.exists(REF_XLSX), reason="reference spreadsheet not mounted (GPU box)")
def test_fixture_matches_spreadsheet(sw):
    """The committed fixture is what the restated `readin_data` produces from the reference's xlsx."""
    from oracle import sw_panel as sp
    dA = sp.readin_data(REF_XLSX, "All")
    dR = sp.readin_data(REF_XLSX, "Real")
    assert np.array_equal(dA["bpdata"], sw["all"], equal_nan=True)
    assert np.array_equal(dR["bpdata"], sw["real"], equal_nan=True)
    assert np.array_equal(dA["inclcode"], sw["inc_all"]) and np.array_equal(dR["inclcode"], sw["inc_real"])
    assert dA["bpnamevec"] == sw["names"]
    assert sp.sample_periods((1959, 1), (2014, 12), 12) == 672 and sp.sample_periods((1959, 1), (2014, 4), 4) == 224


def _check_table2(rows, fn, nrows):
    tr = 1.0 - fn["ssr_static"] / fn["tss"]
    marg = np.concatenate([[tr[0]], np.diff(tr)])
    ah = marg[:-1] / marg[1:]
    for k in range(nrows):
        g = rows[k]
        assert g[0] == k + 1
        assert _shown(tr[k]) == g[1] and _shown(marg[k]) == g[2], (k, tr[k], marg[k], g)
        assert _shown(fn["bn_icp"][k]) == g[3] and _shown(ah[k]) == g[4], (k, fn["bn_icp"][k], ah[k], g)


def test_table2A_real_panel(sw):
    """Stock_Watson.ipynb:572-576 (58 real-activity series, r = 1..5; the driver estimates 1..6)."""
    fn = ao.estimate_factor_numbers(sw["real"], sw["inc_real"], INIT, LAST, 6, with_aw=False, solver="qr")
    assert fn["tss"] == pytest.approx(12700.0, rel=1e-12) and fn["nobs"] == 12700 and fn["T"] == 222
    _check_table2(GOLD["table2A_real"]["rows"], fn, 5)


def test_table2B_2C_full_panel(sw):
    """Stock_Watson.ipynb:619-628 (Bai-Ng etc., r = 1..10 of 11 estimated) and :673-682 (Amengual-Watson)."""
    fn = ao.estimate_factor_numbers(sw["all"], sw["inc_all"], INIT, LAST, 11, with_aw=True, solver="normal")
    assert fn["nobs"] == 29098
    _check_table2(GOLD["table2B_all"]["rows"], fn, 10)
    for i, row in enumerate(GOLD["table2C_aw"]["rows"]):
        assert row[0] == i + 1
        for j in range(10):
            g, x = row[1 + j], fn["aw_icp"][i, j]
            if g is None:
                assert np.isnan(x)
            else:
                assert _shown(x) == g or abs(x - g) <= 5.0001e-4, (i, j, x, g)


def test_solvers_agree(sw):
    """The batched normal-equation sweeps equal the per-regression QR sweeps (the reference's `X\\y`)."""
    a = ao.estimate_factor(sw["real"], sw["inc_real"], INIT, LAST, 4, solver="qr")
    b = ao.estimate_factor(sw["real"], sw["inc_real"], INIT, LAST, 4, solver="normal")
    assert a["iters"] == b["iters"] == 78
    np.testing.assert_allclose(a["ssr_path"], b["ssr_path"], rtol=1e-12)
    np.testing.assert_allclose(a["f"], b["f"], rtol=0, atol=1e-9 * np.abs(a["f"]).max())


@pytest.mark.parametrize("r", [1, 2, 3, 8, 9, 10])
def test_table3_series_r2(sw, r):
    """Stock_Watson.ipynb:991-1017: r2 of `estimate!` for all 207 series, 6 significant digits
    (columns 1-3 and 8-10 and rows 1-13, 196-207 are visible in the saved output)."""
    col = {1: 0, 2: 1, 3: 2, 8: 3, 9: 4, 10: 5}[r]
    o = ao.estimate_factor(sw["all"], sw["inc_all"], INIT, LAST, r, solver="normal", compute_r2_flag=False)
    _, r2, _, _ = ao.estimate_factor_loading(sw["all"], o["factor"], INIT, LAST)
    first, last = GOLD["table3_r2"]["first_rows"], GOLD["table3_r2"]["last_rows"]
    for i, row in enumerate(first):
        assert _sig6(r2[i], row[col]), (i, r2[i], row[col])
    for i, row in enumerate(last):
        k = 207 - len(last) + i
        assert _sig6(r2[k], row[col]), (k, r2[k], row[col])


def test_table5_canonical_correlations(sw):
    """Stock_Watson.ipynb:1250-1261 (tables A, B, O; C needs the stepwise search): pins `estimate_var!`
    residuals of the 8-factor VAR(4) and the factors themselves."""
    bp, names = sw["all"], sw["names"]
    o = ao.estimate_factor(bp, sw["inc_all"], INIT, LAST, 8, solver="normal", compute_r2_flag=False)
    fv = ao.estimate_var(o["factor"], 4, INIT, LAST)
    sets = {"A": ["GDPC96", "PAYEMS", "PCECTPI", "FEDFUNDS"],
            "B": ["GDPC96", "PAYEMS", "PCECTPI", "FEDFUNDS", "NAPMPRI", "WPU0561", "CP90_TBILL", "GS10_TB3M"],
            "O": ["OILPROD_SA", "GLOBAL_ACT", "WPU0561", "GDPC96", "PAYEMS", "PCECTPI", "FEDFUNDS", "TWEXMMTH"]}
    for key, vars_ in sets.items():
        X = np.column_stack([bp[:, names.index(v)] for v in vars_])
        v = ao.estimate_var(X, 4, INIT, LAST)
        ok = ~np.isnan(np.column_stack([X, o["factor"]])).any(axis=1)
        lev = ao.canonical_correlations(X[ok], o["factor"][ok])
        ok = ~np.isnan(np.column_stack([v["resid"], fv["resid"]])).any(axis=1)
        res = ao.canonical_correlations(v["resid"][ok], fv["resid"][ok])
        for x, g in zip(res, GOLD["table5"][key]["resid"]):
            assert _sig6(x, g), (key, x, g)
        for x, g in zip(lev, GOLD["table5"][key]["level"]):
            assert _sig6(x, g), (key, x, g)


def test_var_state_space_matrices(sw):
    """`fill_matrices!` (dfm_functions.ipynb:477-492): companion M, selector Q, G = lower Cholesky of seps;
    IRF recursion (:793-816) equals Q M^t G."""
    o = ao.estimate_factor(sw["real"], sw["inc_real"], INIT, LAST, 4, solver="normal", compute_r2_flag=False)
    v = ao.estimate_var(o["factor"], 4, INIT, LAST)
    assert v["T_used"] == 218 and v["betahat"].shape == (17, 4) and v["M"].shape == (16, 16)
    np.testing.assert_allclose(v["G"][:4] @ v["G"][:4].T, v["seps"], rtol=1e-12)
    assert np.array_equal(v["M"][4:, :12], np.eye(12)) and np.all(v["M"][4:, 12:] == 0)
    irf = ao.impulse_response(v["M"], v["Q"], v["G"], range(4), 12)
    np.testing.assert_allclose(irf[:, 3, 2], v["Q"] @ np.linalg.matrix_power(v["M"], 3) @ v["G"][:, 2], rtol=1e-12)
