"""Sample-rate conversion on the GPU (afec_amd/csrc/afx_resample.hip; SampleAnalyser.cpp:563-607 -> libresample 0.1.3):
files that are not at the analyser's rate go through the C-ABI's LoadSample front end and must leave in the analysis
arena exactly the doubles the oracle's LoadSample leaves -- mix, conversion, rms / peak, normalisation and trim are all
float / double arithmetic in a fixed order, so the comparison is bit for bit.  The oracle's converter is pinned against
the reference's own libresample (tests/test_oracle_resample.py)."""
import numpy as np
import pytest

import afec_amd as afx
from tests import _oracle, _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu


def burst(rng, n, rate, stereo=False, dtype=np.int16):
    t = np.arange(n) / float(rate)
    x = 0.6 * np.sin(2 * np.pi * rng.uniform(60, 3000) * t + rng.uniform(0, 6.28)) + 0.4 * rng.uniform(-1, 1, n) * np.exp(-t * rng.uniform(3, 40))
    x *= rng.uniform(0.2, 0.95) / max(1e-9, np.max(np.abs(x)))
    lead = int(rng.integers(0, max(1, n // 10)))
    x[:lead] = 0.0                                                     # leading silence: the trim runs on the converted samples
    if stereo:
        x = np.stack([x, 0.7 * np.roll(x, 5)], axis=1)
    if dtype == np.int16:
        return np.round(x * 32767).astype(np.int16)
    return x.astype(dtype)


def check_file(batch, i, info, data, channels, rate, what):
    want, winfo = _oracle.load_sample(data, channels, file_rate=rate)
    for k in ("data_offset", "silent_leading", "silent_trailing", "n_samples"):
        assert info[k] == winfo[k], (what, k, info[k], winfo[k])
    assert np.float32(info["peak_value"]) == np.float32(winfo["peak_value"]), what
    assert abs(info["rms_value"] - winfo["rms_value"]) <= 2e-7 * winfo["rms_value"] + 1e-12, what
    plan_frames = (want.size - 2048) // 1024 + 1 if want.size >= 2048 else 0
    kept = min(want.size, (plan_frames - 1) * 1024 + 2048) if plan_frames > 0 else 0
    got = batch.fetch_samples(i, kept)
    np.testing.assert_array_equal(got.view(np.uint64), want[:kept].view(np.uint64), err_msg=what)
    return want


@pytest.mark.parametrize("rate", [48000, 96000, 88200, 22050, 32000, 11025, 8000, 192000, 44099, 44101, 16000])
def test_converted_files_equal_the_oracle_bit_for_bit(rate):
    rng = np.random.default_rng(rate)
    files, meta = [], []
    for n in (1, 2, 40, 3000, 4040, 4041, 9000, rate // 2, rate + 17):
        for stereo in (False, True):
            d = burst(rng, n, rate, stereo)
            files.append((d, 2 if stereo else 1, rate))
            meta.append((d, 2 if stereo else 1, f"{rate} Hz n={n} stereo={stereo}"))
    plan = afx.Plan(max_analysis_ms=0)
    batch, infos = plan.batch_from_raw(files, afx.D_MFCC)
    batch.run()
    res = batch.fetch()
    assert res["buf_status"].tolist() == [0] * len(files)
    for i, (d, ch, what) in enumerate(meta):
        check_file(batch, i, infos[i], d, ch, rate, what)
    batch.close()
    plan.close()


def test_every_sample_type_and_a_mixed_batch():
    """int16 / int24 / int32 / float32 / float64 files at other rates next to files at the analyser's rate in one batch;
    descriptors of the converted files against the oracle pipeline."""
    rng = np.random.default_rng(5)
    x = burst(rng, 30000, 48000, False, np.float64)
    as_i24 = np.frombuffer(b"".join(int(v).to_bytes(4, "little", signed=True)[:3] for v in np.round(x * 8388607).astype(np.int64)), dtype=np.uint8)
    files = [
        (np.round(x * 32767).astype(np.int16), 1, 48000),
        (as_i24.copy(), 1, 48000),
        (np.round(x * 2147483647).astype(np.int64).astype(np.int32), 1, 96000),
        (x.astype(np.float32), 1, 22050),
        (x.copy(), 1, 32000),
        (np.round(x * 32767).astype(np.int16), 1),               # at the analyser's rate
        (np.round(x * 32767).astype(np.int16), 1, 44100),
        (burst(rng, 50000, 48000, True), 2, 48000),
    ]
    plan = afx.Plan()
    mask = afx.D_ALL_LOW_LEVEL | afx.D_STATISTICS
    batch, infos = plan.batch_from_raw(files, mask)
    batch.run()
    res = batch.fetch()
    assert res["buf_status"].tolist() == [0] * len(files)
    ora = Oracle()
    off = res["frame_offset"]
    for i, f in enumerate(files):
        rate = f[2] if len(f) > 2 else 44100
        want = check_file(batch, i, infos[i], f[0], f[1], rate, f"file {i}")
        ref = ora.run(want, cap=True)
        assert off[i + 1] - off[i] == ref.shape[0]
        for field, (a, b) in FIELDS.items():
            if field == "mag":
                continue
            rtol, atol = _tol.GPU_TOL[field]
            _tol.check_gpu(field, res[field][off[i]:off[i + 1]].reshape(ref.shape[0], -1), ref[:, a:b], rtol, atol, what=f"file {i} ")
    np.testing.assert_array_equal(res["mfcc"][off[5]:off[6]], res["mfcc"][off[6]:off[7]])   # rate 0 and rate 44100: the same file
    batch.close()
    plan.close()


def test_rhythm_tracker_sees_the_files_own_rate_and_length():
    """SampleDurationInSeconds / OnsetOffsetInSeconds come from mOriginalSampleRate / mOriginalNumberOfSamples
    (SampleAnalyser.cpp:1001-1004): for a converted file the rate and the length it had on disk."""
    rng = np.random.default_rng(8)
    rate, n = 48000, 3 * 48000
    t = np.arange(n) / rate
    x = np.zeros(n)
    for k in range(6):                                              # a click track at 120 bpm
        s = int(k * 0.5 * rate)
        m = min(n - s, 4000)
        x[s:s + m] += rng.uniform(-1, 1, m) * np.exp(-np.arange(m) / 300.0)
    d = np.round(0.8 * x / np.max(np.abs(x)) * 32767).astype(np.int16)
    plan = afx.Plan()
    batch, infos = plan.batch_from_raw([(d, 1, rate)], afx.D_RHYTHM)
    batch.run()
    got = batch.fetch_rhythm()
    want_pcm, winfo = _oracle.load_sample(d, 1, file_rate=rate)
    want = Oracle().run_rhythm(want_pcm, original_rate=rate, original_samples=n, data_offset=winfo["data_offset"], cap=True)
    np.testing.assert_allclose(got["scalars"][0], want["scalars"], rtol=1e-9, atol=1e-12)
    batch.close()
    plan.close()


def test_a_long_file_and_a_lone_sample():
    """A minute of 96 kHz audio (5.76 M samples in, 2.6 M out: 1 400 input windows of the converter) and files of one
    and two samples."""
    rng = np.random.default_rng(11)
    long_file = burst(rng, 60 * 96000, 96000)
    plan = afx.Plan(max_analysis_ms=0)
    batch, infos = plan.batch_from_raw([(long_file, 1, 96000), (np.array([1234], np.int16), 1, 48000), (np.array([-5, 9000], np.int16), 1, 8000)], afx.D_MFCC)
    batch.run()
    assert batch.fetch()["buf_status"].tolist() == [0, 0, 0]
    check_file(batch, 0, infos[0], long_file, 1, 96000, "long")
    check_file(batch, 1, infos[1], np.array([1234], np.int16), 1, 48000, "one sample")
    check_file(batch, 2, infos[2], np.array([-5, 9000], np.int16), 1, 8000, "two samples")
    batch.close()
    plan.close()


def test_a_file_of_more_than_2_28_converted_samples_is_analysed():
    """A 1.7-hour recording: 136 M samples at 22.05 kHz are 272 M (> 2^28) at the analyser's rate.  Until round 5 such a file
    got AFX_ERR_UNSUPPORTED (a failed row) where the reference converts it and analyses its first 20 s; the bound is 2^30
    now.  LoadSample's normalisation and trim see the WHOLE converted file (the peak sits an hour in, the last audible
    sample at the very end): offsets, peak and the analysed prefix equal the oracle's bit for bit.  The oracle needs
    ~1 minute of CPU for it: AFX_SLOW_TESTS=1 (run by tools/gpu.sh slow; profiles/r05/slow_tests.log)."""
    import os
    if not os.environ.get("AFX_SLOW_TESTS"):
        pytest.skip("AFX_SLOW_TESTS=1: one minute of oracle time")
    rng = np.random.default_rng(21)
    n = (1 << 27) + 2_000_000
    d = np.zeros(n, dtype=np.int16)
    d[3000:3000 + 66150] = burst(rng, 66150, 22050)                    # what the analysed 20 s hold
    d[n // 2:n // 2 + 4000] = np.round(30000 * rng.uniform(-1, 1, 4000)).astype(np.int16)   # the file's peak, an hour in
    d[-3000:-2000] = 2000                                              # the last audible samples
    plan = afx.Plan()
    batch, infos = plan.batch_from_raw([(d, 1, 22050)], afx.D_MFCC)
    batch.run()
    assert batch.fetch()["buf_status"].tolist() == [0]
    assert infos[0]["n_samples"] > 1 << 28
    want, winfo = _oracle.load_sample(d, 1, file_rate=22050)
    for k in ("data_offset", "silent_leading", "silent_trailing", "n_samples"):
        assert infos[0][k] == winfo[k], (k, infos[0][k], winfo[k])
    assert np.float32(infos[0]["peak_value"]) == np.float32(winfo["peak_value"])
    frames = plan.num_frames(winfo["n_samples"])                       # the 20 s cap: 860 frames
    assert frames == 860 and batch.total_frames == frames
    kept = (frames - 1) * 1024 + 2048
    np.testing.assert_array_equal(batch.fetch_samples(0, kept).view(np.uint64), want[:kept].view(np.uint64))
    batch.close()
    plan.close()


def test_real_audio_at_another_rate_through_the_crawler(tmp_path):
    """The reference's own fixture WAVs (tests/golden/wav/) with their headers relabelled 48 kHz and 32 kHz: crawled into
    the database, the per-frame series of the converted audio against the oracle chain (its converter is pinned on the
    reference's libresample), the file properties against the file."""
    import glob
    import os
    import sqlite3
    import struct

    import msgpack

    from tests import _host
    from tests._wav import parse_wav
    wavs = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "wav", "*.wav")) if not os.path.basename(p).startswith("_"))
    assert len(wavs) >= 10
    images, names, want = [], [], {}
    for k, path in enumerate(wavs):
        image = bytearray(open(path, "rb").read())
        channels, rate, bits, frames, payload = parse_wav(bytes(image))
        new_rate = (48000, 32000)[k % 2]
        at = bytes(image).index(b"fmt ") + 8
        struct.pack_into("<II", image, at + 4, new_rate, new_rate * channels * bits // 8)     # nSamplesPerSec, nAvgBytesPerSec
        name = f"relabelled/{new_rate}/{os.path.basename(path)}"
        images.append(bytes(image)); names.append(name)
        want[name] = (np.frombuffer(payload, dtype=np.int16 if bits == 16 else np.uint8), channels, bits, new_rate, frames)
    db = str(tmp_path / "relabelled.db")
    st = _host.crawl(images, names, workers=2, files_per_batch=5, database=db)
    assert st["files"] == len(images) and st["failed"] == 0 and st["skipped_sample_rate"] == 0
    con = sqlite3.connect(db)
    con.row_factory = sqlite3.Row
    ora = Oracle()
    for name, (data, channels, bits, rate, frames) in want.items():
        r = con.execute("SELECT * FROM assets WHERE filename = ?", (name,)).fetchone()
        assert r["status"] == "succeeded" and (r["file_sample_rate_R"], r["file_channel_count_R"], r["file_bit_depth_R"]) == (rate, channels, bits)
        assert abs(r["file_length_R"] - frames / float(rate)) < 1e-6
        mono, info = _oracle.load_sample(data, channels, file_rate=rate)
        ref = ora.run(mono, cap=True)
        for field, col in (("mfcc", "cepstrum_bands_VVR"), ("spectral_centroid", "spectral_centroid_VR"), ("spectrum_bands", "frequency_bands_VVR"),
                           ("amplitude_peak", "amplitude_peak_VR")):
            a, b = FIELDS[field]
            got = np.array(msgpack.unpackb(r[col]), dtype=np.float64).reshape(ref.shape[0], -1)
            _tol.check_gpu(field, got, ref[:, a:b], *_tol.GPU_TOL[field], what=f"{name} {col} ")
    con.close()
