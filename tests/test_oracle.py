"""The oracle (oracle/afx_oracle.c) pinned against the reference.

1. golden fixtures produced by the reference's own compiled objects (tests/golden/make_golden.py);
2. the reference's own known-answer tests for TStatistics
   (Source/Crawler/FeatureExtraction/Test/TestStatistics.cpp:10-115);
3. the survey's known-answer values for a 1 kHz sine;
4. when oracle/_ref/ref_driver is present (build container), a live comparison on fresh inputs.
"""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from tests import _oracle, _tol
from tests._oracle import FIELDS, Oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_DRIVER = os.path.join(_oracle.ROOT, "oracle", "_ref", "ref_driver")

# oracle (radix-2 FFT) vs reference (Ooura split radix): both IEEE double; bins agree to a few
# ulp of the frame's largest bin, so descriptors that take logs of leakage-floor bins agree to
# ~1e-7 relative, everything else to ~1e-10.
ORACLE_RTOL = 1e-6
ORACLE_ATOL = {
    "mfcc": 1e-9, "spectral_flatness": 1e-6, "sub_flatness": 1e-6, "spectral_flux": 1e-9,
    "sub_flux": 1e-7, "spectral_skewness": 1e-12, "spectral_kurtosis": 1e-12,
    "spectral_spread": 1e-9, "spectral_centroid": 1e-9, "sub_contrast": 1e-12,
    "spectral_contrast": 1e-12,
}
# discrete outputs decided by rounding noise on degenerate inputs: an impulse has an exactly flat
# spectrum, so "strict local maximum" (sub_complexity) is a coin toss per bin in ANY implementation
ILL_CONDITIONED = {("impulse", "sub_complexity"), ("impulse", "sub_flux")}


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def golden_names():
    z = np.load(os.path.join(GOLD, "frames.npz"))
    return sorted(k[3:] for k in z.files if k.startswith("in_"))


def test_tables_match_reference(oracle):
    z = np.load(os.path.join(GOLD, "tables.npz"))
    np.testing.assert_array_equal(oracle.window(), z["window"])
    np.testing.assert_array_equal(oracle.mel(), z["mel"])
    assert oracle.first_bin() == 1 and oracle.bin_count() == 738


def test_mel_table_quirks(oracle):
    """filters land on bins 1..358 because of M = N>>1."""
    mel = oracle.mel()
    support = [(int(np.nonzero(r)[0][0]), int(np.nonzero(r)[0][-1])) for r in mel]
    assert support[0] == (1, 3) and support[13] == (225, 358)
    assert mel[0].max() == 0.75 and all(abs(mel[f].max() - 1.0) < 1e-12 for f in range(1, 14))
    assert np.all(mel[:, 359:] == 0.0) and np.all(mel[:, 0] == 0.0)


def test_frame_count_rule(oracle):
    rows = np.load(os.path.join(GOLD, "framecount.npz"))["rows"]
    for n, cap, frames in rows:
        assert oracle.num_frames(int(n), bool(cap)) == int(frames), (n, cap)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(oracle, name):
    z = np.load(os.path.join(GOLD, "frames.npz"))
    x = z["in_" + name].astype(np.float64)
    ref = z["ref_" + name]
    got = oracle.run(x)
    assert got.shape == ref.shape
    _tol.check_mag(got[:, :1024], ref[:, :1024], 1e-13, what=name + " ")
    for field, (a, b) in FIELDS.items():
        if field == "mag" or (name, field) in ILL_CONDITIONED:
            continue
        _tol.check(field, got[:, a:b], ref[:, a:b], ORACLE_RTOL, ORACLE_ATOL.get(field, 1e-15),
                   what=name + " ")


def test_sine_known_answers(oracle):
    """Values printed by the survey's harness built from the reference objects."""
    t = np.arange(2048 + 1024 * 3)
    rec = oracle.run(np.sin(2 * np.pi * 1000 * t / 44100))
    r0 = rec[0]
    assert int(np.argmax(r0[:1024])) == 46
    assert abs(r0[46] - 0.440474) < 1e-6
    assert abs(r0[1024] - (-132.9144003)) < 1e-6
    assert abs(r0[1025] - 11.19628276) < 1e-7
    assert abs(r0[1037] - (-0.3445079189)) < 1e-8
    assert abs(r0[1039] - 45.43424171) < 1e-7
    assert abs(r0[1040] - 1.862741465) < 1e-8
    assert r0[1043] == 2021.0


def test_statistics_known_answers():
    """TestStatistics.cpp:16-115."""
    s = _oracle.stat
    seq = [1, 2, 2, 2, 0, 5, 6]
    assert s("min", seq) == 0 and s("max", seq) == 6
    for q in ([1, 2, 3, 4, 5, 6], [6, 5, 4, 3, 2, 1], [3, 2, 4, 6, 5, 1], [4, 3, 6, 5, 2, 1]):
        assert s("sum", q) == 21
        m = s("mean", q)
        assert abs(s("variance", q, m) - 2.9) < 0.1
        assert s("median", q) == 3
        assert abs(m - 21.0 / 6) < 1e-16
        assert abs(s("geometric_mean", q) - 3) < 1.0
    assert abs(s("centroid", [1, 2, 3, 4, 5, 6]) - 3.0) < 1.0
    assert abs(s("centroid", [1, 1, 1, 1, 6, 8]) - 4.0) < 1.0
    assert abs(s("centroid", [1, 20, 4, 6, 5, 1]) - 2.0) < 1.0
    assert abs(s("centroid", [1, 1, 1, 1, 1, 1]) - 2.5) < 1e-3
    v = [1234567.0]
    assert s("sum", v) == v[0] and s("median", v) == v[0] and s("mean", v) == v[0]
    assert s("variance", v, v[0]) == 0 and s("geometric_mean", v) == v[0]
    assert s("centroid", v) == 0 and s("spread", v, 0.0) == 0
    e = []
    assert s("sum", e) == 0 and s("median", e) == 0 and s("mean", e) == 0
    assert s("variance", e, 0.0) == 0 and s("geometric_mean", e) == 0
    assert s("centroid", e) == 0 and s("spread", e, 0.0) == 0


def test_lin_to_db_edges():
    s = _oracle.stat
    assert s("lin_to_db", 1.0) == 0.0
    assert s("lin_to_db", 0.0) == -200.0
    assert s("lin_to_db", 9e-13) == -200.0          # below (double)1e-12f
    assert abs(s("lin_to_db", 0.5) - 20 * np.log10(0.5)) < 1e-12


def test_calc_statistics_short_series():
    """TStatistics::Calc leaves median/gmean/centroid... untouched for Length<=1 (Statistics.cpp:72-89)."""
    init = np.full(13, 7.0)
    out = _oracle.calc_statistics([3.0], init)
    assert out[0] == 3 and out[1] == 3 and out[3] == 3 and out[5] == 0 and out[11] == 0 and out[12] == 0
    assert out[2] == 7 and out[4] == 7 and out[6] == 7 and out[10] == 7
    out = _oracle.calc_statistics([], init)
    assert out[0] == 0 and out[3] == 0 and out[2] == 7
    out = _oracle.calc_statistics([1, 2, 3, 4, 5, 6])
    assert out[2] == 3 and out[3] == 3.5 and abs(out[11] - 1.0) < 1e-15 and out[12] == 0


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="reference objects only exist in the build container")
def test_oracle_matches_live_reference(oracle):
    rng = np.random.default_rng(7)
    bufs = [rng.uniform(-1, 1, 2048 + 1024 * 5), 0.3 * rng.standard_normal(5000),
            np.sin(2 * np.pi * 3000 * np.arange(9000) / 44100) * np.linspace(1, 0, 9000)]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "i"), os.path.join(d, "o")
        with open(fin, "wb") as f:
            f.write(struct.pack("<q", len(bufs)))
            for b in bufs:
                f.write(struct.pack("<q", b.size))
                f.write(b.astype(np.float64).tobytes())
        subprocess.check_call([REF_DRIVER, "frames", fin, fout, "0"])
        raw = open(fout, "rb").read()
    ref = np.frombuffer(raw[8:], dtype=np.float64).reshape(-1, _oracle.RECORD)
    got = np.concatenate([oracle.run(b) for b in bufs])
    assert got.shape == ref.shape
    _tol.check_mag(got[:, :1024], ref[:, :1024], 1e-13)
    for field, (a, b) in FIELDS.items():
        if field != "mag":
            _tol.check(field, got[:, a:b], ref[:, a:b], ORACLE_RTOL, ORACLE_ATOL.get(field, 1e-15))


# ---- stateful neighbours --------------------------------------------------------

# an impulse has an exactly flat spectrum: the whitened spectrum's local maxima are rounding noise
NEIGH_ILL_CONDITIONED = {("impulse", "spectral_complexity")}


def neighbour_names():
    z = np.load(os.path.join(GOLD, "neighbours.npz"))
    return sorted(k[4:] for k in z.files if k.startswith("ref_"))


def neighbour_input(name):
    z = np.load(os.path.join(GOLD, "neighbours.npz"))
    if "in_" + name in z.files:
        return z["in_" + name].astype(np.float64)
    return np.load(os.path.join(GOLD, "frames.npz"))["in_" + name].astype(np.float64)


@pytest.mark.parametrize("name", neighbour_names())
def test_oracle_neighbours_match_reference_golden(oracle, name):
    """whitening / peaks / silence / envelope / yinfast pitch / autocorrelation against the reference's
    own aubio, TEnvelopeDetector, TAutocorrelation and LibXtract objects."""
    ref = np.load(os.path.join(GOLD, "neighbours.npz"))["ref_" + name]
    got = oracle.run_neighbours(neighbour_input(name))
    assert got.shape[0] == ref.shape[0]
    if ref.shape[1] > 11:
        _tol.check_mag(got[:, 11:], ref[:, 11:], 1e-11, what=name + " whitened ")
    for field, col in _oracle.NEIGH_FIELDS.items():
        if (name, field) in NEIGH_ILL_CONDITIONED:
            continue
        _tol.check(field, got[:, col], ref[:, col], 1e-9, 1e-12, what=name + " ")


def test_peaks_known_answer_from_reference_test():
    """TestStatistics.cpp:16-33."""
    seq = [1, 2, 2, 2, 0, 5, 6]
    assert _oracle.peaks(seq, 0) == [(2, 2.0), (6, 6.0)]
    assert _oracle.peaks(seq, 2) == [(6, 6.0)]
    assert _oracle.peaks([1, 2], 0) == []
    assert _oracle.peaks([3, 1, 2, 1, 5], 0) == [(0, 3.0), (2, 2.0), (4, 5.0)]


def test_neighbour_descriptors_that_are_identically_zero(oracle):
    """LibXtract reads partial frequencies from the cleared upper half of the spectrum buffer (SURVEY 8a note):
    inharmonicity and tristimulus are 0 for every frame of every golden signal in the reference's output."""
    z = np.load(os.path.join(GOLD, "neighbours.npz"))
    for k in z.files:
        if k.startswith("ref_"):
            assert not z[k][:, 7:11].any(), k


# ---- LoadSample front end -------------------------------------------------------

def load_names():
    z = np.load(os.path.join(GOLD, "load.npz"))
    return sorted(k[4:] for k in z.files if k.startswith("raw_"))


@pytest.mark.parametrize("name", load_names())
def test_oracle_load_sample_matches_reference_golden(name):
    """conversion, mono mix, peak / rms, normalisation, -48 dB trim and padding against `ref_driver load`
    (the reference's TSampleConverter / TMathT / TAudioMath arithmetic around the restated flow): bit-exact"""
    z = np.load(os.path.join(GOLD, "load.npz"))
    got, info = _oracle.load_sample(z["raw_" + name], int(z["channels_" + name]))
    off, lead, trail, n = z["info_" + name].tolist()
    assert (info["data_offset"], info["silent_leading"], info["silent_trailing"], info["n_samples"]) == (off, lead, trail, n)
    assert np.float32(info["peak_value"]) == z["peakrms_" + name][0]
    assert np.float32(info["rms_value"]) == z["peakrms_" + name][1]
    np.testing.assert_array_equal(got, z["data_" + name])


# ---- effective length (per file) ----------------------------------------------------------------

def efflen_input(z, name):
    """inputs of efflen.npz live there, or in frames.npz / neighbours.npz when they are the shared golden signals"""
    if "in_" + name in z.files:
        return z["in_" + name]
    return _shared_golden_input(name)


def _shared_golden_input(name):
    for fn in ("frames.npz", "neighbours.npz"):
        zz = np.load(os.path.join(GOLD, fn))
        if "in_" + name in zz.files:
            return zz["in_" + name]
    raise KeyError(name)


def test_oracle_effective_length_matches_reference_golden(oracle):
    """CalcEffectiveLength (SampleAnalyser.cpp:1715-1755) against `ref_driver efflen`: exact"""
    z = np.load(os.path.join(GOLD, "efflen.npz"))
    names = sorted(k[4:] for k in z.files if k.startswith("ref_"))
    assert len(names) >= 10
    for name in names:
        got = oracle.effective_length(efflen_input(z, name).astype(np.float64))
        np.testing.assert_array_equal(got, z["ref_" + name], err_msg=name)
    # the three floors are nested
    for name in names:
        r = z["ref_" + name]
        assert r[0] >= r[1] >= r[2] >= 0.0
