"""GPU parity of the half-wave STFT + MFCC kernel (afec_amd/csrc/afx_frames32.hip): the MFCC-only class with
f32 PCM, forced on for every batch size with afx_plan_desc.frame_kernel = AFX_FRAME_KERNEL_HALFWAVE, against the
reference goldens, the oracle and the 64-lane kernel (AFX_FRAME_KERNEL_WAVE64).  Covers what the work queue and the two-chunks-per-wave walk can get wrong:
odd chunk counts, chunks of different lengths in the two halves of a wave, one-frame buffers, repeated launches
on one batch (the queue counter is never reset) and workspace reuse between batches."""
import os

import numpy as np
import pytest

import afec_amd as afx
from tests import _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_plan(mode):
    """mode 0: the 64-lane kernel for every batch, 1: by batch size (the default), 2: the half-wave kernel for every batch"""
    return afx.Plan(max_analysis_ms=0, frame_kernel={0: afx.FRAME_KERNEL_WAVE64, 1: afx.FRAME_KERNEL_AUTO,
                                                      2: afx.FRAME_KERNEL_HALFWAVE}[mode])


@pytest.fixture(scope="module")
def forced():
    p = make_plan(2)
    yield p
    p.close()


@pytest.fixture(scope="module")
def lanes64():
    p = make_plan(0)
    yield p
    p.close()


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def golden_names():
    z = np.load(os.path.join(GOLD, "frames.npz"))
    return sorted(k[3:] for k in z.files if k.startswith("in_"))


@pytest.mark.parametrize("name", golden_names())
def test_golden_mfcc(forced, name):
    z = np.load(os.path.join(GOLD, "frames.npz"))
    x, ref = z["in_" + name], z["ref_" + name]
    res = forced.extract([x.astype(np.float32)], afx.D_MFCC)
    a, b = FIELDS["mfcc"]
    # goldens were generated from double input; the signals are exactly representable in float32 or the
    # float32 rounding is part of both sides only when the test converts: compare against the oracle on the
    # float32 signal instead when the cast changes the input
    if np.array_equal(x.astype(np.float32).astype(np.float64), x):
        _tol.check_gpu("mfcc", res["mfcc"], ref[:, a:b], *_tol.GPU_TOL["mfcc"], what=name + " ")
    else:
        o = Oracle().run(x.astype(np.float32).astype(np.float64))
        _tol.check_gpu("mfcc", res["mfcc"], o[:, a:b], *_tol.GPU_TOL["mfcc"], what=name + " (oracle) ")


def test_ragged_batch_matches_oracle_and_the_64_lane_kernel(forced, lanes64, oracle):
    rng = np.random.default_rng(5)
    # 0 and sub-frame buffers, one-frame buffers, lengths that leave chunks of different sizes in the two halves
    lens = [0, 100, 2048, 2048, 3072, 2048 + 1024 * 2, 2048 + 1024 * 33, 2048 + 1024 * 64, 2048 + 1024 * 97, 2047,
            2048 + 1024 * 31, 2048]
    bufs = [rng.uniform(-1, 1, n).astype(np.float32) for n in lens]
    bufs[6] = (0.5 * np.sin(2 * np.pi * 440.0 * np.arange(lens[6]) / 44100)).astype(np.float32)   # tonal
    bufs[7][:] = 0.0                                                                             # silence: log floor
    got = forced.extract(bufs, afx.D_MFCC)
    ref64 = lanes64.extract(bufs, afx.D_MFCC)
    assert got["frame_offset"].tolist() == ref64["frame_offset"].tolist()
    assert got["buf_status"].tolist() == ref64["buf_status"].tolist()
    # two FFT factorisations of the same double arithmetic
    _tol.check("mfcc", got["mfcc"], ref64["mfcc"], 1e-9, 1e-9, what="half-wave vs 64-lane ")
    a, b = FIELDS["mfcc"]
    off = got["frame_offset"]
    for i, x in enumerate(bufs):
        if off[i + 1] == off[i]:
            continue
        o = oracle.run(x.astype(np.float64))
        _tol.check_gpu("mfcc", got["mfcc"][off[i]:off[i + 1]], o[:, a:b], *_tol.GPU_TOL["mfcc"], what=f"buffer {i} ")


def test_repeated_launches_and_workspace_reuse(forced):
    rng = np.random.default_rng(9)
    a = [rng.uniform(-1, 1, 2048 + 1024 * 150).astype(np.float32) for _ in range(5)]
    b = [rng.uniform(-1, 1, 2048 + 1024 * 17).astype(np.float32) for _ in range(3)]
    first = forced.extract(a, afx.D_MFCC)["mfcc"].copy()
    batch = forced.batch(a, afx.D_MFCC)
    for _ in range(4):            # the work-queue counter advances from launch to launch
        batch.run()
    batch.sync()
    np.testing.assert_array_equal(batch.fetch()["mfcc"], first)
    batch.close()
    other = forced.extract(b, afx.D_MFCC)["mfcc"].copy()    # the pooled workspace (and its counter) changes hands
    np.testing.assert_array_equal(forced.extract(a, afx.D_MFCC)["mfcc"], first)
    np.testing.assert_array_equal(forced.extract(b, afx.D_MFCC)["mfcc"], other)


def test_default_plan_picks_the_half_wave_kernel_for_large_batches(forced, oracle):
    """Above the size threshold a default plan takes the same path (bitwise the same results)."""
    rng = np.random.default_rng(21)
    bufs = [rng.uniform(-1, 1, 2048 + 1024 * 2999).astype(np.float32) for _ in range(24)]   # 72 000 frames
    auto = make_plan(1)
    try:
        got = auto.extract(bufs, afx.D_MFCC)["mfcc"]
    finally:
        auto.close()
    np.testing.assert_array_equal(got, forced.extract(bufs, afx.D_MFCC)["mfcc"])
    a, b = FIELDS["mfcc"]
    o = oracle.run(bufs[3][:2048 + 1024 * 40].astype(np.float64))
    _tol.check_gpu("mfcc", got[3 * 3000:3 * 3000 + 41], o[:, a:b], *_tol.GPU_TOL["mfcc"], what="large batch ")


# ---- the statistics class of the half-wave kernel: MFCC + spectral rms / centroid / spread / skewness / kurtosis /
#      rolloff / flatness (bins 0..767, sums reduced per half, closed forms in stats32_finish_kernel) ----
STAT_FIELDS = ["mfcc", "spectral_rms", "spectral_centroid", "spectral_spread", "spectral_skewness", "spectral_kurtosis",
               "spectral_rolloff", "spectral_flatness"]
STAT_MASK = afx.D_MFCC | afx.D_SPECTRAL_STATS & ~afx.D_SPECTRAL_FLUX


def check_stats(got, ref_rows, what):
    for field in STAT_FIELDS:
        if field not in got:
            continue
        a, b = FIELDS[field]
        _tol.check_gpu(field, got[field].reshape(ref_rows.shape[0], -1), ref_rows[:, a:b], *_tol.GPU_TOL[field], what=what)


@pytest.mark.parametrize("name", golden_names())
def test_golden_statistics_class(forced, name):
    z = np.load(os.path.join(GOLD, "frames.npz"))
    x, ref = z["in_" + name], z["ref_" + name]
    x32 = x.astype(np.float32)
    if not np.array_equal(x32.astype(np.float64), x):
        ref = Oracle().run(x32.astype(np.float64))
    check_stats(forced.extract([x32], STAT_MASK), ref, name + " ")


def test_statistics_class_ragged_batch_and_partial_masks(forced, lanes64, oracle):
    rng = np.random.default_rng(15)
    lens = [0, 2048, 3072, 2048 + 1024 * 5, 2048 + 1024 * 33, 2048 + 1024 * 64, 2047, 2048 + 1024 * 31, 2048 + 1024 * 70]
    bufs = [rng.uniform(-1, 1, n).astype(np.float32) for n in lens]
    t = np.arange(lens[4])
    bufs[4] = (0.5 * np.sin(2 * np.pi * 1000.0 * t / 44100) + 0.2 * np.sin(2 * np.pi * 7000.0 * t / 44100)).astype(np.float32)
    bufs[5][:] = 0.0                                      # silence: every sum is 0, flatness / rolloff special cases
    bufs[8] = (rng.standard_normal(lens[8]) * np.exp(-np.arange(lens[8]) / 9000.0)).astype(np.float32)
    star = afx.D_MFCC | afx.D_SPECTRAL_RMS | afx.D_SPECTRAL_CENTROID | afx.D_SPECTRAL_SPREAD | afx.D_SPECTRAL_ROLLOFF | afx.D_SPECTRAL_FLATNESS
    for mask in (STAT_MASK, star, afx.D_MFCC | afx.D_SPECTRAL_KURTOSIS, afx.D_MFCC | afx.D_SPECTRAL_ROLLOFF):
        got = forced.extract(bufs, mask)
        ref64 = lanes64.extract(bufs, mask)
        assert got["frame_offset"].tolist() == ref64["frame_offset"].tolist()
        off = got["frame_offset"]
        for i, x in enumerate(bufs):
            if off[i + 1] == off[i]:
                continue
            o = oracle.run(x.astype(np.float64))
            check_stats({k: v[off[i]:off[i + 1]] for k, v in got.items() if k in STAT_FIELDS}, o, f"mask {mask:#x} buffer {i} ")
        # the two layouts agree far inside the tolerance (rolloff exactly)
        if "spectral_rolloff" in got:
            assert np.array_equal(got["spectral_rolloff"], ref64["spectral_rolloff"])
        if "spectral_centroid" in got:
            _tol.check("spectral_centroid", got["spectral_centroid"], ref64["spectral_centroid"], 1e-9, 1e-9, what="half-wave vs 64-lane ")


def test_statistics_class_repeated_launches(forced):
    rng = np.random.default_rng(29)
    a = [rng.uniform(-1, 1, 2048 + 1024 * 150).astype(np.float32) for _ in range(5)]
    first = forced.extract(a, STAT_MASK)
    batch = forced.batch(a, STAT_MASK)
    for _ in range(3):
        batch.run()
    batch.sync()
    again = batch.fetch()
    batch.close()
    for f in STAT_FIELDS:
        np.testing.assert_array_equal(again[f], first[f])


# ---- the full class of the half-wave kernel: the statistics class + every magnitude stored (all mirrored blocks) + the
#      amplitude of the hop; flux, the 28 spectrum bands and the sub-band descriptors from bands_kernel ----
FULL_FIELDS = [f for f in FIELDS if f != "mag"]


def check_full(got, ref_rows, what, rows=None):
    for field in FULL_FIELDS:
        if field not in got:
            continue
        a, b = FIELDS[field]
        g = got[field] if rows is None else got[field][rows[0]:rows[1]]
        _tol.check_gpu(field, g.reshape(ref_rows.shape[0], -1), ref_rows[:, a:b], *_tol.GPU_TOL[field], what=what)


@pytest.mark.parametrize("name", golden_names())
def test_golden_full_class(forced, name):
    z = np.load(os.path.join(GOLD, "frames.npz"))
    x, ref = z["in_" + name], z["ref_" + name]
    x32 = x.astype(np.float32)
    if not np.array_equal(x32.astype(np.float64), x):
        ref = Oracle().run(x32.astype(np.float64))
    got = forced.extract([x32], afx.D_ALL_LOW_LEVEL | afx.D_MAGNITUDE)
    # ill-conditioned by construction (tests/test_gpu_parity.py): flux and the sub-band complexity /
    # flux of an exactly flat spectrum
    if name == "impulse":
        got = {k: v for k, v in got.items() if k not in ("sub_complexity", "sub_flux", "spectral_flux")}
    check_full(got, ref, name + " ")
    _tol.check_mag(got["magnitude"].reshape(ref.shape[0], 1024), ref[:, :1024], what=name + " ")


def test_full_class_ragged_batch_against_oracle_and_the_64_lane_kernel(forced, lanes64, oracle):
    rng = np.random.default_rng(35)
    lens = [0, 2048, 3072, 2048 + 1024 * 5, 2048 + 1024 * 33, 2048 + 1024 * 64, 2047, 2048 + 1024 * 31, 2048 + 1024 * 70]
    bufs = [rng.uniform(-1, 1, n).astype(np.float32) for n in lens]
    t = np.arange(lens[4])
    bufs[4] = (0.5 * np.sin(2 * np.pi * 1000.0 * t / 44100) + 0.2 * np.sin(2 * np.pi * 7000.0 * t / 44100)).astype(np.float32)
    bufs[5][:] = 0.0                                      # silence: every magnitude flushed to exactly 0
    bufs[8] = (rng.standard_normal(lens[8]) * np.exp(-np.arange(lens[8]) / 9000.0)).astype(np.float32)
    masks = (afx.D_ALL_LOW_LEVEL | afx.D_MAGNITUDE, afx.D_ALL_LOW_LEVEL, afx.D_MFCC | afx.D_SPECTRUM_BANDS,
             afx.D_MFCC | afx.D_AMPLITUDE_PEAK | afx.D_AMPLITUDE_RMS, afx.D_MFCC | afx.D_SPECTRAL_FLUX | afx.D_SPECTRAL_ROLLOFF)
    for mask in masks:
        got = forced.extract(bufs, mask)
        ref64 = lanes64.extract(bufs, mask)
        assert got["frame_offset"].tolist() == ref64["frame_offset"].tolist()
        off = got["frame_offset"]
        for i, x in enumerate(bufs):
            if off[i + 1] == off[i]:
                continue
            o = oracle.run(x.astype(np.float64))
            check_full(got, o, f"mask {mask:#x} buffer {i} ", rows=(off[i], off[i + 1]))
            if "magnitude" in got:
                _tol.check_mag(got["magnitude"][off[i]:off[i + 1]].reshape(-1, 1024), o[:, :1024], what=f"buffer {i} ")
        if "magnitude" in got:
            silent = got["magnitude"][off[5]:off[6]]
            assert silent.size > 0 and np.all(silent == 0.0)           # flushed, not sqrt(DBL_MIN)
        for field in ("spectral_rolloff", "sub_complexity", "amplitude_peak"):
            if field in got:
                assert np.array_equal(got[field], ref64[field]), field
        if "spectrum_bands" in got:
            # two FFT factorisations: equal to rounding relative to the frame's energy (leakage-floor bands carry the noise)
            floor = 1e-24 * got["spectrum_bands"].reshape(-1, 28).sum(axis=1, keepdims=True)
            g, r = got["spectrum_bands"].reshape(-1, 28), ref64["spectrum_bands"].reshape(-1, 28)
            assert np.all(np.abs(g - r) <= 1e-9 * np.abs(r) + floor + 1e-300)


def test_full_class_repeated_launches(forced):
    rng = np.random.default_rng(39)
    a = [rng.uniform(-1, 1, 2048 + 1024 * 150).astype(np.float32) for _ in range(5)]
    first = forced.extract(a, afx.D_ALL_LOW_LEVEL)
    batch = forced.batch(a, afx.D_ALL_LOW_LEVEL)
    for _ in range(3):
        batch.run()
    batch.sync()
    again = batch.fetch()
    batch.close()
    for f in FULL_FIELDS:
        np.testing.assert_array_equal(again[f], first[f])
