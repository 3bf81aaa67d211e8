"""Stock-Watson (2016) panel: NumPy restatement of the reference's data ingestion.

TEST INFRASTRUCTURE ONLY (see oracle/kalman_oracle.py header).  Used by
tests/golden/make_sw_fixture.py to turn the reference's spreadsheet
(`/root/reference/data/hom_fac_1.xlsx`, present only in the build container) into the
small committed fixture `tests/golden/sw_panel.npz`, and by tests/test_oracle_sw.py to
check that fixture against the spreadsheet whenever the reference is mounted.

Follows `readin_functions.jl` line by line in *behaviour* (cited per function); the
spreadsheet is read with the standard library only (zipfile + a streaming XML parse),
because neither ExcelReaders/xlrd nor openpyxl exist in this image.

Missing cells are NaN throughout (the reference uses `Union{Missing,Float64}`).
"""
from __future__ import annotations

import re
import xml.etree.ElementTree as ET
import zipfile
from typing import Dict, List, Tuple

import numpy as np

_NS = "{http://schemas.openxmlformats.org/spreadsheetml/2006/main}"
_CELL = re.compile(r"([A-Z]+)(\d+)")


def _col_index(letters: str) -> int:
    n = 0
    for ch in letters:
        n = n * 26 + (ord(ch) - 64)
    return n - 1


def read_xlsx_sheet(path: str, sheet_name: str) -> List[List[object]]:
    """Dense row-major grid of a sheet: float for numeric cells, str for text, None for empty/error.

    Stands in for `readxlsheet("data/hom_fac_1.xlsx", sheet)` (readin_functions.jl:204-205)."""
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
        rid = None
        for s in wb.iter(_NS + "sheet"):
            if s.get("name") == sheet_name:
                rid = s.get("{http://schemas.openxmlformats.org/officeDocument/2006/relationships}id")
        if rid is None:
            raise KeyError(sheet_name)
        target = None
        for r in rels:
            if r.get("Id") == rid:
                target = r.get("Target")
        shared: List[str] = []
        for si in ET.fromstring(z.read("xl/sharedStrings.xml")).iter(_NS + "si"):
            shared.append("".join(t.text or "" for t in si.iter(_NS + "t")))
        root = ET.fromstring(z.read("xl/" + target))
    cells: Dict[Tuple[int, int], object] = {}
    nrow = ncol = 0
    for c in root.iter(_NS + "c"):
        m = _CELL.fullmatch(c.get("r"))
        col, row = _col_index(m.group(1)), int(m.group(2)) - 1
        t = c.get("t", "n")
        v = c.find(_NS + "v")
        val: object = None
        if t == "s" and v is not None:
            val = shared[int(v.text)]
        elif t in ("str", "inlineStr"):
            if v is not None:
                val = v.text
            else:
                val = "".join(x.text or "" for x in c.iter(_NS + "t"))
        elif t == "n" and v is not None and v.text not in (None, ""):
            val = float(v.text)
        elif t == "b" and v is not None:
            val = float(v.text)
        if val is None:
            continue
        cells[(row, col)] = val
        nrow, ncol = max(nrow, row + 1), max(ncol, col + 1)
    grid: List[List[object]] = [[None] * ncol for _ in range(nrow)]
    for (r_, c_), v in cells.items():
        grid[r_][c_] = v
    return grid


# ----------------------------------------------------------------------------- dates
def _excel_serial_to_ym(serial: float) -> Tuple[int, int]:
    """Excel 1900-system serial -> (year, month).  21551 = 1959-01-01."""
    import datetime as dt
    d = dt.date(1899, 12, 30) + dt.timedelta(days=int(round(serial)))
    return d.year, d.month


def sample_periods(init: Tuple[int, int], last: Tuple[int, int], per_year: int) -> int:
    """`MonthlyData`/`QuarterlyData` constructors (readin_functions.jl:29-36)."""
    return per_year * (last[0] - init[0] - 1) + last[1] + (per_year - init[1] + 1)


# ----------------------------------------------------------------------------- transforms
def transform(x: np.ndarray, tcode: int) -> np.ndarray:
    """readin_functions.jl:104-115 (1 level, 2 diff, 3 second diff, 4 log, 5 dlog, 6 d2log)."""
    if tcode in (4, 5, 6):
        with np.errstate(invalid="ignore", divide="ignore"):
            x = np.log(x)
        tcode -= 3
    if tcode == 1:
        return x.copy()
    out = np.full_like(x, np.nan)
    if tcode == 2:
        out[1:] = x[1:] - x[:-1]
    elif tcode == 3:
        out[2:] = x[2:] - 2.0 * x[1:-1] + x[:-2]
    else:
        raise ValueError(tcode)
    return out


def adjust_outlier(x: np.ndarray, ocode: int, io_method: int = 4) -> None:
    """In place.  readin_functions.jl:127-149 (threshold 4.5 IQR for code 1, 3 IQR for code 2) and
    :189-198 for io_method 4: every flagged cell becomes the median of itself and the 5 preceding cells
    (missing skipped), evaluated sequentially in place so earlier replacements feed later windows."""
    if ocode == 0:
        return
    thr = {1: 4.5, 2: 3.0}[ocode]
    obs = x[~np.isnan(x)]
    zm = np.median(obs)
    iqr = np.quantile(obs, 0.75) - np.quantile(obs, 0.25)     # Julia default = type 7 = NumPy default
    if not iqr >= 1e-6:
        raise RuntimeError("error in adjusting outlier")
    with np.errstate(invalid="ignore"):
        flagged = np.abs(x - zm) > thr * iqr                   # NaN -> False (missing & ... filtered, :192)
    if io_method != 4:
        raise NotImplementedError("the reference's driver only uses io_method=4 (readin_functions.jl:200-203)")
    for i in np.flatnonzero(flagged):
        w = x[max(0, i - 5): i + 1]
        x[i] = np.median(w[~np.isnan(w)])


def bi_weight_filter(y: np.ndarray, weight: float) -> np.ndarray:
    """Tukey-biweight local mean over observed points only (readin_functions.jl:335-348)."""
    T = y.shape[0]
    obs = ~np.isnan(y)
    trend = np.full(T, np.nan)
    idx = np.arange(1, T + 1, dtype=float)
    yo = y[obs]
    for t in np.flatnonzero(obs):
        dt_ = (idx - (t + 1)) / weight
        w = 15.0 / 16.0 * (1.0 - dt_ ** 2) ** 2
        w[np.abs(dt_) >= 1.0] = 0.0
        wo = w[obs]
        wo = wo / wo.sum()
        trend[t] = wo @ yo
    return trend


# ----------------------------------------------------------------------------- one sheet
def _readin_sheet(path: str, monthly: bool, datatype: str, nobs: int, ns: int,
                  correct_outlier: bool = True, io_method: int = 4, ndesc: int = 2,
                  cat_include=(1, 2, 3, 5)):
    """`readin_monthly_data` (readin_functions.jl:206-253) for either sheet."""
    ncodes = 6 if monthly else 5                                           # :200-203
    grid = read_xlsx_sheet(path, "Monthly" if monthly else "Quarterly")
    top = 1 + ndesc + ncodes
    main = [row[1: ns + 1] + [None] * (ns + 1 - len(row)) for row in grid[: top + nobs]]   # :215
    serials = [grid[top + t][0] for t in range(nobs)]                      # :216
    ym = [_excel_serial_to_ym(s) for s in serials]
    names = [str(v).upper() for v in main[0][:ns]]                         # :259 / :273
    off = 1 if monthly else 0                                              # monthly sheet has AggCode row (:262)
    ints = lambda r: np.array([int(v) for v in main[r][:ns]])
    tcode, defcode, outcode, inclcode = ints(3 + off), ints(4 + off), ints(5 + off), ints(6 + off)
    catcode = np.array([float(v) for v in main[7 + off][:ns]])
    dat = np.full((nobs, ns), np.nan)
    for t in range(nobs):
        row = main[top + t]
        for j in range(ns):
            v = row[j]
            if isinstance(v, float):                                       # :225 non-Float64 -> missing
                dat[t, j] = v
    # deflators (:285-301)
    col = lambda nm: dat[:, names.index(nm)].copy()
    if monthly:
        price_def, price_lfe, price_gdp = col("PCEPI"), col("PCEPILFE"), None
        j = names.index("GLOBAL_ACT")                                      # Kilian index (:306-313), n-1 sd
        o = ~np.isnan(dat[:, j])
        v = dat[o, j]
        dat[o, j] = (v - v.mean()) / v.std(ddof=1)
    else:
        price_def, price_lfe, price_gdp = col("PCECTPI"), col("JCXFE"), col("GDPCTPI")
    if datatype == "Real":                                                 # :254-256
        used = (inclcode != 0) & np.isin(np.floor(catcode), cat_include)
    elif datatype == "All":
        used = inclcode != 0
    else:
        raise ValueError(datatype)
    data = dat[:, used].copy()
    dcode, ocode, tc = defcode[used], outcode[used], tcode[used]
    for i in range(data.shape[1]):                                         # :244-245, deflate_series! :40-76
        if dcode[i] == 1:
            data[:, i] /= price_def
        elif dcode[i] == 2:
            data[:, i] /= price_lfe
        elif dcode[i] == 3:
            data[:, i] /= price_gdp
    if monthly:                                                            # :83-96 mean of the quarter's months
        q = [(y, (mth + 2) // 3) for y, mth in ym]
        uq = sorted(set(q))
        qarr = np.array([uq.index(k) for k in q])
        data_q = np.stack([data[qarr == k].mean(axis=0) for k in range(len(uq))])   # NaN if any month missing
        date_q = uq
    else:
        data_q, date_q = data, [(y, (mth + 2) // 3) for y, mth in ym]
    raw = data_q.copy()
    for i in range(data_q.shape[1]):                                       # :248
        data_q[:, i] = transform(data_q[:, i], int(tc[i]))
    noa = data_q.copy()
    if correct_outlier:                                                    # :250
        for i in range(data_q.shape[1]):
            adjust_outlier(data_q[:, i], int(ocode[i]), io_method)
    nm = [n for n, u in zip(names, used) if u]
    return data_q, raw, noa, date_q, catcode[used], inclcode[used], nm


def readin_data(path: str, datatype: str, bw_weight: float = 100.0,
                m_init=(1959, 1), m_last=(2014, 12), m_ns=148,
                q_init=(1959, 1), q_last=(2014, 4), q_ns=85):
    """`readin_data(md, qd, BiWeight(100), :Real|:All)` (readin_functions.jl:355-382) with the
    driver's constants (`Stock_Watson.ipynb:143-144, 160, 180`)."""
    md_nobs = sample_periods(m_init, m_last, 12)
    qd_nobs = sample_periods(q_init, q_last, 4)
    dm, rawm, noam, datem, catm, incm, nmm = _readin_sheet(path, True, datatype, md_nobs, m_ns)
    dq, rawq, noaq, dateq, catq, incq, nmq = _readin_sheet(path, False, datatype, qd_nobs, q_ns)
    if datem != dateq:
        raise RuntimeError("inconsistent sample size for monthly and quarterly data")
    cat = np.concatenate([catm, catq])
    order = np.argsort(cat, kind="stable")                                 # sortperm is stable (:366)
    bp = np.hstack([dm, dq])[:, order]
    unfiltered = bp.copy()
    trend = np.full_like(bp, np.nan)
    for i in range(bp.shape[1]):                                           # :317-324
        trend[:, i] = bi_weight_filter(bp[:, i], bw_weight)
        bp[:, i] = bp[:, i] - trend[:, i]
    names = (nmm + nmq)
    return dict(bpdata=bp, bpdata_unfiltered=unfiltered, bpdata_trend=trend,
                bpdata_raw=np.hstack([rawm, rawq])[:, order], bpdata_noa=np.hstack([noam, noaq])[:, order],
                bpcatcode=cat[order], inclcode=np.concatenate([incm, incq])[order],
                bpnamevec=[names[i] for i in order], calds=dateq,
                calvec=np.array([y + (q - 1) / 4 for y, q in dateq]))
