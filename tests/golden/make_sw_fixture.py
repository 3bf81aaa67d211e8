"""Build tests/golden/sw_panel.npz from the reference's spreadsheet (run in the build container only).

    python tests/golden/make_sw_fixture.py [/root/reference]

The fixture is the `:All` Stock-Watson panel exactly as `readin_data(md, qd, BiWeight(100), :All)` returns
it (readin_functions.jl:355-382; constants Stock_Watson.ipynb:143-144, 180): `bpdata` 224 x 207 (NaN =
missing), `inclcode`, `bpcatcode`, `bpnamevec`, `calvec`.  The `:Real` panel of Stock_Watson.ipynb:160 is
the column subset floor(bpcatcode) in {1,2,3,5} (readin_functions.jl:254) - asserted here and in
tests/test_oracle_sw.py.  `/root/reference` does not exist on the GPU box, so tests read this file.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sw_panel as sp  # noqa: E402

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
xlsx = os.path.join(ref, "data", "hom_fac_1.xlsx")
dA = sp.readin_data(xlsx, "All")
dR = sp.readin_data(xlsx, "Real")
keep = np.isin(np.floor(dA["bpcatcode"]), [1, 2, 3, 5])
assert np.array_equal(dA["bpdata"][:, keep], dR["bpdata"], equal_nan=True)
assert np.array_equal(dA["inclcode"][keep], dR["inclcode"])
assert dA["bpdata"].shape == (224, 207) and dR["bpdata"].shape == (224, 86)
out = os.path.join(ROOT, "tests", "golden", "sw_panel.npz")
np.savez_compressed(out, bpdata=dA["bpdata"], inclcode=dA["inclcode"].astype(np.int64),
                    bpcatcode=dA["bpcatcode"], bpnamevec=np.array(dA["bpnamevec"]), calvec=dA["calvec"])
print(out, os.path.getsize(out), "bytes")
