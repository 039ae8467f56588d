"""the LoadSample front end on the GPU (afx_batch_create_from_raw) against the
oracle's restatement of SampleAnalyser.cpp:484-718.  Integer / float conversion, mix-down, trim
offsets and the normalised samples must be bit-exact; rms is a sum in a different order."""
import numpy as np
import pytest

import afec_amd as afx
from tests import _oracle, _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu


def pack24(x):
    v = np.clip(np.round(x * 8388607.0), -8388608, 8388607).astype(np.int32)
    b = np.zeros((v.size, 3), dtype=np.uint8)
    b[:, 0], b[:, 1], b[:, 2] = v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF
    return b.reshape(-1)


def make_files():
    rng = np.random.default_rng(31)
    t = np.arange(30000)
    tone = 0.3 * np.sin(2 * np.pi * 440 * t / 44100) * np.exp(-t / 9000.0)
    lead = np.concatenate([np.zeros(2205), tone, np.zeros(4000)])
    stereo = np.stack([lead, 0.8 * np.roll(lead, 7)], axis=1)
    files = [
        ((lead * 20000).astype(np.int16), 1),                               # mono 16-bit, leading + trailing silence
        ((stereo * 30000).astype(np.int16), 2),                             # stereo 16-bit
        (pack24(0.5 * rng.uniform(-1, 1, 5000)), 1),                        # mono 24-bit, no silence
        (rng.uniform(-1.2, 1.2, (7000, 2)).astype(np.float32), 2),          # float stereo with clipping
        (np.zeros(3000, dtype=np.int16), 1),                                # digital silence
        ((0.6 * np.sin(2 * np.pi * 100 * np.arange(700) / 44100) * 32767).astype(np.int16), 1),  # shorter than a frame
        (np.concatenate([np.zeros(100), [0.5], np.zeros(100)]).astype(np.float32), 1),           # one loud sample
        ((rng.uniform(-1, 1, (4097, 6)) * 9000).astype(np.int16), 6),       # 5.1
    ]
    return files


def test_load_front_end_matches_oracle():
    files = make_files()
    plan = afx.Plan()
    mask = afx.D_MFCC | afx.D_SPECTRAL_CENTROID | afx.D_AMPLITUDE_PEAK
    batch, infos = plan.batch_from_raw(files, mask)
    batch.run()
    res = batch.fetch()
    ora = Oracle()
    row = 0
    for i, (data, ch) in enumerate(files):
        want, winfo = _oracle.load_sample(data, ch)
        got_info = infos[i]
        for k in ("data_offset", "silent_leading", "silent_trailing", "n_samples"):
            assert got_info[k] == winfo[k], (i, k, got_info[k], winfo[k])
        assert got_info["peak_value"] == winfo["peak_value"], i
        assert abs(got_info["rms_value"] - winfo["rms_value"]) <= 2e-7 * max(1e-30, winfo["rms_value"]) + 1e-12, i
        nf = plan.num_frames(len(want))
        kept = (nf - 1) * 1024 + 2048 if nf > 0 else 0
        got = batch.fetch_samples(i, kept)
        np.testing.assert_array_equal(got, want[:kept], err_msg=f"file {i}")
        ref = ora.run(want, cap=True)
        assert ref.shape[0] == nf
        a, b = FIELDS["mfcc"]
        _tol.check_gpu("mfcc", res["mfcc"][row:row + nf], ref[:, a:b], *_tol.GPU_TOL["mfcc"], what=f"file {i} ")
        a, b = FIELDS["amplitude_peak"]
        np.testing.assert_array_equal(res["amplitude_peak"][row:row + nf], ref[:, a])
        row += nf
    assert row == batch.total_frames
    batch.close()
    plan.close()


def test_load_front_end_matches_reference_golden():
    """afx_batch_create_from_raw against tests/golden/load.npz (`ref_driver load`: the reference's converters and
    maths around the restated LoadSample flow): samples, offsets and peak bit-exact; the rms sum is re-associated"""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "load.npz"))
    names = sorted(k[4:] for k in z.files if k.startswith("raw_"))
    files = [(z["raw_" + n], int(z["channels_" + n])) for n in names]
    plan = afx.Plan()
    batch, infos = plan.batch_from_raw(files, afx.D_MFCC)
    for i, n in enumerate(names):
        off, lead, trail, size = z["info_" + n].tolist()
        got = infos[i]
        assert (got["data_offset"], got["silent_leading"], got["silent_trailing"], got["n_samples"]) == (off, lead, trail, size), n
        assert np.float32(got["peak_value"]) == z["peakrms_" + n][0], n
        assert abs(got["rms_value"] - z["peakrms_" + n][1]) <= 2e-7 * z["peakrms_" + n][1] + 1e-12, n
        nf = plan.num_frames(size)
        kept = (nf - 1) * 1024 + 2048 if nf > 0 else 0
        np.testing.assert_array_equal(batch.fetch_samples(i, kept), z["data_" + n][:kept], err_msg=n)
    batch.close()
    plan.close()


def test_load_front_end_rejects_bad_files_individually():
    rng = np.random.default_rng(32)
    good = (rng.uniform(-1, 1, 5000) * 20000).astype(np.int16)
    plan = afx.Plan()
    batch, infos = plan.batch_from_raw([(good, 1), (good, 9), (good, 1, -5), (good, 1)], afx.D_MFCC)
    batch.run()
    res = batch.fetch()
    assert res["buf_status"].tolist() == [0, -6, -6, 0]     # 9 channels, a negative sampling rate: bad buffers
    assert res["frame_offset"].tolist() == [0, 4, 4, 4, 8]   # 5000 audible + 1024 end pad -> 4 frames
    np.testing.assert_array_equal(res["mfcc"][:4], res["mfcc"][4:])
    batch.close()
    plan.close()


def test_pads_are_zero_in_a_reused_workspace():
    """The analysis arena is not cleared between batches (pooled workspaces): load_write_kernel itself writes the zeros of
    the start pad, the end pad and the slot's slack.  A batch of loud files first fills the arena, then batches of short /
    silent / heavily trimmed files reuse it: samples (pads included) and descriptors must equal the oracle's."""
    rng = np.random.default_rng(77)
    plan = afx.Plan()
    mask = afx.D_MFCC | afx.D_AMPLITUDE_PEAK | afx.D_AUTO_CORRELATION | afx.D_EFFECTIVE_LENGTH | afx.D_RHYTHM
    loud = [((rng.uniform(-1, 1, 50000) * 30000).astype(np.int16), 1) for _ in range(12)]
    ora = Oracle()
    for round_ in range(3):
        b0, _ = plan.batch_from_raw(loud, mask)
        b0.run(); b0.fetch(); b0.close()
        files = []
        for k in range(12):
            n = int(rng.choice([300, 700, 1500, 2047, 2049, 3000, 5000, 9000]))
            x = np.zeros(n)
            kind = (k + round_) % 4
            if kind == 0:
                x[n // 3:n // 3 + 40] = rng.uniform(-0.8, 0.8, 40)                      # a click in silence: long pads
            elif kind == 1:
                x[:] = 0.5 * np.sin(2 * np.pi * 300 * np.arange(n) / 44100)
            elif kind == 2:
                x[-5:] = 0.4                                                             # audible only at the very end
            files.append(((x * 32767).astype(np.int16), 1))                              # kind 3: digital silence
        batch, infos = plan.batch_from_raw(files, mask)
        batch.run()
        res = batch.fetch()
        row = 0
        for i, (data, ch) in enumerate(files):
            want, _ = _oracle.load_sample(data, ch)
            got = batch.fetch_samples(i, len(want))
            np.testing.assert_array_equal(got, want[:len(got)], err_msg=f"round {round_} file {i}")
            nf = plan.num_frames(len(want))
            ref = ora.run(want, cap=True)
            a, b = FIELDS["amplitude_peak"]
            np.testing.assert_array_equal(res["amplitude_peak"][row:row + nf], ref[:, a])
            a, b = FIELDS["mfcc"]
            _tol.check_gpu("mfcc", res["mfcc"][row:row + nf], ref[:, a:b], *_tol.GPU_TOL["mfcc"], what=f"round {round_} file {i} ")
            row += nf
        batch.close()
    plan.close()


def test_upload_reads_only_the_callers_bytes():
    """A single buffer (or buffers that lie back to back) goes to the device in one transfer; that transfer must end
    with the last file's last sample, not at the 16-byte slot behind it: here the samples end at the last byte of a
    mapping whose next page is inaccessible (a file image, an mmap-backed array)."""
    import ctypes
    import mmap
    page = mmap.PAGESIZE
    m = mmap.mmap(-1, 4 * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(base + 3 * page, page, 0) == 0          # PROT_NONE behind the data
    n = (3 * page - 6) // 2                                      # an odd number of samples: the slot rounds up by 6 bytes... and more
    n -= 1                                                       # ends 2 bytes... keep it not a multiple of 16
    view = np.frombuffer(m, dtype=np.int16, count=3 * page // 2)
    rng = np.random.default_rng(3)
    x = np.round(8000 * np.sin(2 * np.pi * 440 * np.arange(n) / 44100) + rng.uniform(-200, 200, n)).astype(np.int16)
    tail = view[view.size - n:]                                  # the last n samples of the accessible pages
    tail[:] = x
    assert (tail.ctypes.data + 2 * n) == base + 3 * page and (2 * n) % 16 != 0
    plan = afx.Plan(device=0)
    try:
        batch, infos = plan.batch_from_raw([(tail, 1)], afx.D_MFCC)
        batch.run()
        got = batch.fetch()["mfcc"]
        batch.close()
        b2, _ = plan.batch_from_raw([(x.copy(), 1)], afx.D_MFCC)
        b2.run()
        want = b2.fetch()["mfcc"]
        b2.close()
        np.testing.assert_array_equal(got, want)
    finally:
        plan.close()
        del view, tail
        libc.mprotect(base + 3 * page, page, 3)


def test_the_arena_behind_loadsample_holds_floats_and_one_scale_per_file():
    """LoadSample's output is (double)float_sample * FinalScaling (SampleAnalyser.cpp:710-718): the batch keeps the
    float signal and the scale (afx_batch_info.pcm_kind = 2, four bytes a sample), afx_batch_fetch_samples forms the
    doubles on demand, and a batch made of exactly those doubles (the reference's mData, AFX_PCM_F64) gives the same
    descriptors -- every kernel forms the same products when it loads a sample."""
    files = make_files()
    plan = afx.Plan()
    mask = afx.D_ALL_PER_FRAME
    batch, infos = plan.batch_from_raw(files, mask)
    info = batch.info()
    assert info["pcm_kind"] == 2
    wants = [_oracle.load_sample(data, ch)[0] for data, ch in files]
    kept = []
    for w in wants:
        nf = plan.num_frames(w.size)
        kept.append(min(w.size, (nf - 1) * 1024 + 2048 + 64) if nf > 0 else 0)     # (+ 64: the autocorrelation's second search)
    assert info["arena_bytes"] == 4 * sum((k + 3) // 4 * 4 for k in kept)
    batch.run()
    res = batch.fetch()
    for i, w in enumerate(wants):
        np.testing.assert_array_equal(batch.fetch_samples(i, kept[i]), w[:kept[i]], err_msg=f"file {i}")
    batch.close()
    again = plan.batch(wants, mask)
    assert again.info()["pcm_kind"] == afx.PCM_F64 and again.info()["arena_bytes"] == 2 * info["arena_bytes"]
    again.run()
    res64 = again.fetch()
    for field in res:
        if field in ("frame_offset", "buf_status"):
            continue
        np.testing.assert_array_equal(res[field], res64[field], err_msg=field)
    again.close()
    plan.close()


def test_large_batches_behind_loadsample_take_the_half_wave_frame_kernel():
    """A batch of the crawl's size class (BASELINE configs[2]: a thousand two-second files) with every per-frame descriptor
    runs its STFT on the half-wave kernel's magnitude class (MFCC + the stored spectrum; the statistics, flux and band
    descriptors come from bands_kernel), a small one on the 64-lane kernel -- the crossover is measured
    (tools/x_batchsize.py, profiles/r04); the results of the two agree to rounding (both against the oracle elsewhere)."""
    rng = np.random.default_rng(12)
    pool = [np.round(8000 * rng.uniform(-1, 1, 88200)).astype(np.int16) for _ in range(8)]
    plan = afx.Plan()
    mask = afx.D_ALL_PER_FRAME | afx.D_STATISTICS
    big, _ = plan.batch_from_raw([(pool[i % 8], 1) for i in range(600)], mask)
    small, _ = plan.batch_from_raw([(pool[i % 8], 1) for i in range(8)], mask)
    bi, si = big.info(), small.info()
    assert bi["pcm_kind"] == si["pcm_kind"] == 2
    assert bi["frame_kernel"] == afx.FRAME_KERNEL_HALFWAVE and bi["feature_class"] == 4
    assert si["frame_kernel"] == afx.FRAME_KERNEL_WAVE64 and si["feature_class"] == 2
    big.run(); small.run()
    rb, rs = big.fetch(), small.fetch()
    nf = small.total_frames
    assert big.total_frames == 75 * nf
    for field in rs:
        if field in ("frame_offset", "buf_status"):
            continue
        a, b = rb[field][:nf].astype(np.float64), rs[field].astype(np.float64)
        scale = np.maximum(np.abs(b), 1e-9)
        if field in ("sub_complexity", "spectral_complexity", "sub_flux", "spectral_flux", "f0", "failsafe_f0", "f0_confidence"):
            continue          # discrete / ill-conditioned on noise: compared with the oracle's tolerances in their own tests
        assert np.max(np.abs(a - b) / scale) < 1e-6, field
    big.close(); small.close(); plan.close()


@pytest.mark.parametrize("kernel", ["wave64", "halfwave"])
def test_a_files_rows_do_not_depend_on_the_batch_it_was_analysed_in(kernel):
    """SampleAnalyser.cpp:368-408 analyses a file on its own; here a file's frames are cut into chunks whose length the
    planner picks for the whole batch, and the kernels carry state from frame to frame inside a chunk (the previous
    spectrum and its band sums, groups of frames finished together).  Every per-frame value and statistic of a file must
    be the same bit pattern whatever the batch size, the file's position in it and the chunk length were."""
    rng = np.random.default_rng(77)
    t = np.arange(66150)
    pool = [np.round(9000 * rng.uniform(-1, 1, 66150)).astype(np.int16),
            np.round(12000 * np.sin(2 * np.pi * 220 * t / 44100) * np.exp(-t / 20000.0)).astype(np.int16),
            np.round(3000 * rng.standard_normal(66150) * (t % 11025 < 2000)).astype(np.int16),
            np.concatenate([np.zeros(20000), np.round(15000 * rng.uniform(-1, 1, 46150))]).astype(np.int16)]
    plan = afx.Plan(frame_kernel=afx.FRAME_KERNEL_WAVE64 if kernel == "wave64" else afx.FRAME_KERNEL_HALFWAVE)
    mask = afx.D_ALL_PER_FRAME | afx.D_STATISTICS
    results, chunk_frames = [], set()
    # (800: from 768 buffers on the whitening kernels take a whole file of <= 128 frames as ONE chunk and follow_kernel is
    # not launched -- afx_batch_plan.cpp, cut_whitening_chunks -- where the smaller batches walk chunks of K frames from
    # follower states follow_kernel left: the same bits either way)
    for n_files in (4, 9, 150, 601, 800):
        b, _ = plan.batch_from_raw([(pool[i % 4], 1) for i in range(n_files)], mask)
        chunk_frames.add(b.info()["chunk_frames"])
        b.run()
        res, st = b.fetch(), b.fetch_statistics()
        off = res["frame_offset"]
        rows = {}
        for i in (0, 1, 2, 3, n_files - 4, n_files - 3, n_files - 2, n_files - 1, n_files // 2):
            rows[i] = ({k: v[off[i]:off[i + 1]] for k, v in res.items() if k not in ("frame_offset", "buf_status")},
                       {k: v[i] for k, v in st.items()})
        results.append(rows)
        b.close()
    assert len(chunk_frames) >= 2, chunk_frames       # the planner did cut the files differently
    want = {c: results[0][c] for c in range(4)}
    for rows in results:
        for i, (frames, stats) in rows.items():
            for k, v in frames.items():
                np.testing.assert_array_equal(v, want[i % 4][0][k], err_msg=f"{k} of file {i}")
            for k, v in stats.items():
                np.testing.assert_array_equal(v, want[i % 4][1][k], err_msg=f"statistics {k} of file {i}")
    plan.close()


def test_the_magnitude_class_without_the_upper_spectrum_leaves_the_same_values():
    """frames32_kernel<6> (round 6): for the spectral set without the loop's neighbours nobody reads bins 769..1023, and
    from 131 072 frames on the half-wave magnitude class neither stores nor roots them.  What it does store and every
    descriptor derived from it must be the bit patterns of class 4: the same files in a batch below the threshold (class
    4) and above it (class 6), half-wave layout pinned, every value and statistic of the first files compared."""
    rng = np.random.default_rng(606)
    t = np.arange(44100)
    pool = [np.round(9000 * rng.uniform(-1, 1, 44100)).astype(np.int16),
            np.round(12000 * np.sin(2 * np.pi * 330 * t / 44100) * np.exp(-t / 15000.0)).astype(np.int16),
            np.round(4000 * rng.standard_normal(44100) * (t % 9000 < 2500)).astype(np.int16),
            np.concatenate([np.zeros(15000), np.round(15000 * rng.uniform(-1, 1, 29100))]).astype(np.int16)]
    plan = afx.Plan(frame_kernel=afx.FRAME_KERNEL_HALFWAVE)
    mask = afx.D_ALL_LOW_LEVEL | afx.D_STATISTICS
    results = []
    for n_files in (40, 4200):                      # 28-37 frames per file after the trim: ~1 400 and ~150 000 frames
        b, _ = plan.batch_from_raw([(pool[i % 4], 1) for i in range(n_files)], mask)
        assert b.info()["frame_kernel"] == afx.FRAME_KERNEL_HALFWAVE and b.info()["feature_class"] == 4
        assert (b.total_frames >= 131072) == (n_files == 4200), b.total_frames
        b.run()
        res, st = b.fetch(), b.fetch_statistics()
        off = res["frame_offset"]
        results.append(({k: v[off[0]:off[8]] for k, v in res.items() if k not in ("frame_offset", "buf_status")},
                        {k: v[:8] for k, v in st.items()}))
        b.close()
    plan.close()
    for part in (0, 1):
        for k, v in results[0][part].items():
            assert np.array_equal(np.asarray(v).view(np.uint8), np.asarray(results[1][part][k]).view(np.uint8)), k
