"""CPU: the host restatement of the device panel generator (oracle/synth_oracle.py) -- Philox4x32-10 against the
known-answer vectors of the Random123 distribution (kat_vectors), and the DGP built on it against SURVEY §8(d)."""
import numpy as np

from oracle import kalman_oracle as ko
from oracle import synth_oracle as so

KAT = [  # counter, key, expected (Random123 kat_vectors: "philox4x32 10 ...")
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for c, k, want in KAT:
        got = so.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want
    # vectorised call = elementwise calls
    cs = np.array([c for c, _, _ in KAT], dtype=np.uint32)
    ks = np.array([k for _, k, _ in KAT], dtype=np.uint32)
    got = so.philox4x32_10(cs, ks)
    for row, (_, _, want) in zip(got, KAT):
        assert tuple(int(v) for v in row) == want


def test_device_dgp_restatement_is_the_survey_dgp():
    N, T, r = 60, 400, 4
    x, p = so.synth_replicate_device(20160415, 5, N, T, r)
    np.testing.assert_allclose(x.mean(0), 0.0, atol=1e-12)
    np.testing.assert_allclose(x.std(0), 1.0, rtol=1e-12)
    np.testing.assert_allclose(np.diag(p["A"]), np.linspace(0.5, 0.9, r))
    np.testing.assert_allclose(p["Q"], np.eye(r) - p["A"] @ p["A"].T)
    # standardised series: idiosyncratic + common variance = 1 (up to sampling error); noise share = R / (R + |lam|^2)
    tot = p["R"] + (p["Lam"] ** 2).sum(1)
    assert 0.8 < np.median(tot) < 1.2 and 0.05 < np.median(p["R"] / tot) < 0.6
    # the smoother at the DGP's own parameters explains the panel about as well as for the host DGP of the parity subset
    ll = ko.kfs_pass(x, **p, lag_one=False)["loglik"]
    xh, ph = ko.synth_replicate(5, N, T, r)
    llh = ko.kfs_pass(xh, **ph, lag_one=False)["loglik"]
    assert abs(ll - llh) < 0.05 * abs(llh)
    # distinct replicates / seeds are distinct streams; missing cells are iid with the requested probability
    x2, _ = so.synth_replicate_device(20160415, 6, N, T, r)
    assert np.abs(np.corrcoef(x[:, 0], x2[:, 0])[0, 1]) < 0.2
    xm, _ = so.synth_replicate_device(20160415, 5, N, T, r, missing=0.1)
    frac = np.isnan(xm).mean()
    assert 0.09 < frac < 0.11
    np.testing.assert_array_equal(xm[~np.isnan(xm)], x[~np.isnan(xm)])
