"""Transcribe the numbers saved in the reference driver's cell outputs into tests/golden/notebook_goldens.json.

    python tests/golden/make_notebook_goldens.py [/root/reference]

These are the only known-answer values the reference holds: the notebook was executed by
its author (Julia 1.0.2) and the printed tables were saved with it.  Raw-line citations of
`Stock_Watson.ipynb` are recorded next to every block.  Run in the build container only (the GPU box has no
/root/reference); the JSON is committed.
"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
raw = open(os.path.join(ref, "Stock_Watson.ipynb"), encoding="utf-8").read().split("\n")
ansi = re.compile(r"\\u001b\[\d+m")


def line(n):                      # raw 1-based line -> the JSON string on it, ANSI colour codes removed
    s = raw[n - 1].strip().rstrip(",")
    return ansi.sub("", json.loads(s))


def table_rows(first, last):      # Millboard table rows: | k | c1 | c2 | ... |
    out = []
    for n in range(first, last + 1):
        cells = [c.strip() for c in line(n).strip().strip("|").split("|")]
        out.append([float(c) if c != "NaN" else None for c in cells[1:]])
    return out


def matrix_rows(first, last):     # Julia matrix display rows (skips the ellipsis row)
    out = []
    for n in range(first, last + 1):
        toks = [t for t in line(n).split() if t not in ("…", "⋮", "⋱")]
        if toks:
            out.append([float(t) for t in toks])
    return out


def vec(n):
    return [float(t) for t in line(n).strip().strip("[]").split()]


g = {
    "_source": "Stock_Watson.ipynb saved outputs (Julia 1.0.2); raw-line numbers of the .ipynb file",
    "table2A_real": {"lines": "572-576", "cols": ["nfac", "trace_r2", "marg_r2", "bn_icp2", "ah_er"],
                     "rows": table_rows(572, 576)},
    "table2B_all": {"lines": "619-628", "cols": ["nfac", "trace_r2", "marg_r2", "bn_icp2", "ah_er"],
                    "rows": table_rows(619, 628)},
    "table2C_aw": {"lines": "673-682", "cols": ["ndyn"] + [f"static{k}" for k in range(1, 11)],
                   "rows": table_rows(673, 682)},
    "table3_r2": {"lines": "992-1017", "shape": [207, 10], "visible_cols": [0, 1, 2, 7, 8, 9],
                  "first_rows": matrix_rows(992, 1004), "last_rows": matrix_rows(1006, 1017)},
    "table4": {"lines": "1132-1168", "chow_qlr_r4": matrix_rows(1132, 1134), "chow_qlr_r8": matrix_rows(1144, 1146),
               "cor_r4": matrix_rows(1156, 1157), "cor_r8": matrix_rows(1167, 1168)},
    "table5": {"lines": "1250-1261",
               "B": {"resid": vec(1251), "level": vec(1252)}, "A": {"resid": vec(1254), "level": vec(1255)},
               "C": {"resid": vec(1257), "level": vec(1258)}, "O": {"resid": vec(1260), "level": vec(1261)}},
    "dims": {"quarterly": [224, 85], "lastperiod": 224, "lines": "134, 213"},
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "notebook_goldens.json")
json.dump(g, open(out, "w"), indent=1)
print(out)
