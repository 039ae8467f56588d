"""GPU parity: the HIP path, called through the C-ABI, against the reference goldens and the
oracle.  Bar (BASELINE.json north_star): every descriptor within 1e-4 relative of the CPU
reference; tolerances and their absolute floors are in tests/_tol.py."""
import os

import numpy as np
import pytest

import afec_amd as afx
from tests import _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MASK_FRAMES = afx.D_ALL_LOW_LEVEL
ILL_CONDITIONED = {("impulse", "sub_complexity"), ("impulse", "sub_flux"), ("impulse", "spectral_flux")}


@pytest.fixture(scope="module")
def plan():
    p = afx.Plan(max_analysis_ms=0)
    yield p
    p.close()


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def golden_names():
    z = np.load(os.path.join(GOLD, "frames.npz"))
    return sorted(k[3:] for k in z.files if k.startswith("in_"))


def compare(res, ref, mask, what, skip=()):
    for field, (a, b) in FIELDS.items():
        if field == "mag":
            if mask & afx.D_MAGNITUDE:
                _tol.check_mag(res["magnitude"], ref[:, a:b], 1e-12, what=what)
            continue
        if field not in res or field in skip:
            continue
        rtol, atol = _tol.GPU_TOL[field]
        got = res[field].reshape(ref.shape[0], -1)
        _tol.check_gpu(field, got, ref[:, a:b], rtol, atol, what=what)


def test_tables_bit_exact(plan):
    z = np.load(os.path.join(GOLD, "tables.npz"))
    np.testing.assert_array_equal(plan.window(), z["window"])
    np.testing.assert_array_equal(plan.mel_table(), z["mel"])
    assert plan.bin_range() == (1, 738)


def test_frame_count_rule():
    rows = np.load(os.path.join(GOLD, "framecount.npz"))["rows"]
    capped, free = afx.Plan(max_analysis_ms=20000), afx.Plan(max_analysis_ms=0)
    for n, cap, frames in rows:
        assert (capped if cap else free).num_frames(int(n)) == int(frames), (n, cap)
    capped.close(); free.close()


@pytest.mark.parametrize("name", golden_names())
def test_golden_f64(plan, name):
    z = np.load(os.path.join(GOLD, "frames.npz"))
    x, ref = z["in_" + name], z["ref_" + name]
    mask = MASK_FRAMES | afx.D_MAGNITUDE
    res = plan.extract([x], mask)
    assert res["frame_offset"].tolist() == [0, ref.shape[0]]
    skip = {f for (n, f) in ILL_CONDITIONED if n == name}
    compare(res, ref, mask, name + " ", skip)


def test_ragged_batch_matches_oracle(plan, oracle):
    rng = np.random.default_rng(11)
    lens = [0, 100, 2047, 2048, 3071, 3072, 5000, 20000, 2048 + 1024 * 40]
    bufs = [rng.uniform(-1, 1, n).astype(np.float32) for n in lens]
    res = plan.extract(bufs, MASK_FRAMES)
    want = [oracle.run(b.astype(np.float64)) for b in bufs]
    off = np.cumsum([0] + [w.shape[0] for w in want])
    assert res["frame_offset"].tolist() == off.tolist()
    assert res["buf_status"].tolist() == [0] * len(bufs)
    compare(res, np.concatenate(want), MASK_FRAMES, "ragged ")


def test_f64_pcm_input(plan, oracle):
    rng = np.random.default_rng(12)
    x = 0.25 * rng.standard_normal(2048 + 1024 * 9)
    res = plan.extract([x], MASK_FRAMES)
    compare(res, oracle.run(x), MASK_FRAMES, "f64 pcm ")


def test_analysis_cap(oracle):
    p = afx.Plan(max_analysis_ms=20000)
    rng = np.random.default_rng(13)
    x = rng.uniform(-1, 1, 900000).astype(np.float32)
    res = p.extract([x], afx.D_MFCC | afx.D_SPECTRAL_CENTROID)
    assert res["mfcc"].shape == (860, 14)
    want = oracle.run(x.astype(np.float64), cap=True)
    idx = [0, 1, 31, 32, 33, 500, 858, 859]
    _tol.check_gpu("mfcc", res["mfcc"][idx], want[idx, 1024:1038], *_tol.GPU_TOL["mfcc"])
    p.close()


def test_bad_buffer_does_not_fail_the_batch(plan, oracle):
    import ctypes
    from afec_amd import capi
    rng = np.random.default_rng(14)
    good = rng.uniform(-1, 1, 4096).astype(np.float32)
    arr, keep = capi._pack_bufs([good, good, good])
    arr[1].pcm = None            # NULL pointer with a positive length
    out, res = capi._alloc_out(afx.D_MFCC, 6, 3)
    st = plan.L.afx_extract_batch(plan.h, arr, 3, afx.D_MFCC, ctypes.byref(out))
    assert st == 0
    assert res["buf_status"].tolist() == [0, -6, 0]
    assert res["frame_offset"].tolist() == [0, 3, 3, 6]
    want = oracle.run(good.astype(np.float64))[:, 1024:1038]
    _tol.check_gpu("mfcc", res["mfcc"], np.concatenate([want, want]), *_tol.GPU_TOL["mfcc"])


def test_chunking_invariance_bitwise(plan):
    """Frame f of a long buffer == frame 0 of the 2048-sample slice starting at f*1024."""
    rng = np.random.default_rng(15)
    x = rng.uniform(-1, 1, 2048 + 1024 * 99).astype(np.float32)
    mask = afx.D_MFCC | afx.D_SPECTRAL_STATS & ~afx.D_SPECTRAL_FLUX | afx.D_SPECTRUM_BANDS
    long = plan.extract([x], mask)
    picks = [0, 1, 5, 31, 32, 63, 64, 99]
    short = plan.extract([x[f * 1024: f * 1024 + 2048] for f in picks], mask)
    for k in ("mfcc", "spectral_centroid", "spectral_rolloff", "spectrum_bands", "spectral_flatness"):
        np.testing.assert_array_equal(long[k][picks], short[k])


def test_c2_full_size_against_oracle_sample(plan, oracle):
    """BASELINE config C2: 10 000 frames of U(-1,1) from std::mt19937(1234) (SURVEY 8d's generator), MFCC only."""
    from afec_amd import hostlib
    x = hostlib.fill_uniform_mt19937(10241024, 1234)
    assert x.dtype == np.float32 and abs(float(x[0]) + 0.6169611) < 1e-6 and -1.0 <= x.min() and x.max() < 1.0
    res = plan.extract([x], afx.D_C2)
    assert res["mfcc"].shape == (10000, 14)
    assert np.all(np.isfinite(res["mfcc"]))
    for f0 in (0, 4990, 9936):
        want = oracle.run_mfcc(x[f0 * 1024: f0 * 1024 + 2048 + 63 * 1024].astype(np.float64))
        _tol.check_gpu("mfcc", res["mfcc"][f0:f0 + 64], want, *_tol.GPU_TOL["mfcc"], what=f"C2 f0={f0} ")


def test_resident_batch_rerun_is_deterministic(plan):
    rng = np.random.default_rng(16)
    bufs = [rng.uniform(-1, 1, 2048 + 1024 * 50).astype(np.float32) for _ in range(4)]
    b = plan.batch(bufs, MASK_FRAMES)
    b.run(); b.sync()
    first = b.fetch()
    ms = b.run_timed(3)
    assert ms > 0
    second = b.fetch()
    for k in first:
        np.testing.assert_array_equal(first[k], second[k])
    b.close()
