"""The streaming, sharded host driver (afec_amd/host/Crawler.cpp) end to end on the GPU: WAV file images in (C3 /
C4-shaped synthetic files, every WAV sample type, and files the reader must reject) -> WAV reader -> page-locked
staging -> LoadSample + every per-frame descriptor + statistics on the GPU -> raw record download -> ONE writer ->
the reference's sqlite `assets` table, read back with Python's sqlite3 + msgpack and compared with the oracle."""
import os
import sqlite3

import numpy as np
import pytest

import afec_amd as afx
from tests import _host, _oracle, _tol
from tests._oracle import FIELDS, Oracle
from tests._wav import wav_bytes

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "load_wav.npz")
# series name in the C-ABI -> descriptor base name in the database (SampleDescriptors.cpp:150-205)
DB_NAME = {"spectral_rms": "spectral_rms", "spectral_centroid": "spectral_centroid", "spectral_rolloff": "spectral_rolloff",
           "spectral_flatness": "spectral_flatness", "amplitude_peak": "amplitude_peak", "amplitude_rms": "amplitude_rms",
           "mfcc": "cepstrum_bands", "spectrum_bands": "frequency_bands", "sub_rms": "spectral_rms_bands",
           "sub_contrast": "spectral_contrast_bands"}


def synth(rng, seconds, stereo):
    n = int(44100 * seconds)
    t = np.arange(n) / 44100.0
    x = np.zeros(n)
    for _ in range(int(rng.integers(1, 4))):
        x += rng.uniform(0.2, 0.6) * np.sin(2 * np.pi * rng.uniform(110.0, 4000.0) * t + rng.uniform(0, 6.28))
    x += rng.uniform(0.2, 0.8) * rng.uniform(-1, 1, n) * np.exp(-t / rng.uniform(0.05, 0.5))
    x[:2205] = 0.0
    x *= rng.uniform(0.1, 0.9) / np.max(np.abs(x))
    if stereo:
        return np.stack([x, 0.8 * np.roll(x, 7)], axis=1), 2
    return x, 1


def make_crawl(n_files=120):
    rng = np.random.default_rng(91)
    z = np.load(GOLD)
    images, names, decoded = [], [], {}
    for i in range(n_files):
        x, ch = synth(rng, 2.0 if i % 2 == 0 else 1.0, stereo=(i % 2 == 1))
        pcm = np.round(x * 32767).astype(np.int16)
        names.append(f"Synth/{'stereo' if ch == 2 else 'mono'}_{i:04d}.wav")
        images.append(wav_bytes(pcm, ch, 16, extra_chunks=(i % 7 == 0)))
        decoded[names[-1]] = (pcm, ch)
    for k in ("u8_mono", "i24_mono", "i32_stereo", "f32_mono", "f64_stereo"):
        names.append(f"Formats/{k}.wav")
        images.append(z["wav_" + k].tobytes())
    names.append("Broken/_Not A Wavefile.wav")
    images.append(b"RIFF....this is not a wave file" * 8)
    # not at the analyser's rate: converted on the GPU as the reference converts them on the CPU (SampleAnalyser.cpp:563-607)
    for rate, stereo in ((48000, True), (22050, False), (96000, False)):
        x, ch = synth(rng, 1.5, stereo)
        pcm = np.round(x * 32767).astype(np.int16)
        names.append(f"Rates/{rate}_{'stereo' if stereo else 'mono'}.wav")
        images.append(wav_bytes(pcm, ch, 16, rate=rate))
        decoded[names[-1]] = (pcm, ch, rate)
    return images, names, decoded


def test_crawl_into_the_descriptor_database(tmp_path):
    import msgpack
    images, names, decoded = make_crawl()
    db = str(tmp_path / "afec-ll.db")
    st = _host.crawl(images, names, devices=(0,), workers=3, files_per_batch=16, database=db)
    assert st["files"] == len(images) and st["failed"] == 1 and st["skipped_sample_rate"] == 0
    assert st["files_per_device"] == [len(images)]
    con = sqlite3.connect(db)
    con.row_factory = sqlite3.Row
    assert con.execute("PRAGMA user_version").fetchone()[0] == 2
    rows = {r["filename"]: r for r in con.execute("SELECT * FROM assets")}
    assert len(rows) == len(images)
    assert rows["Broken/_Not A Wavefile.wav"]["status"] == "error: Sample failed to load: Not a valid WAV file."   # SampleAnalyser.cpp:372-387
    assert sum(1 for r in rows.values() if r["status"] != "succeeded") == 1          # cf. UnitTests.cpp:338-350
    z = np.load(GOLD)
    ora = Oracle()
    rng = np.random.default_rng(5)
    picks = list(rng.choice(120, 6, replace=False)) + [names.index(n) for n in names if n.startswith("Rates/")]
    for i in picks:
        name = names[i]
        r = rows[name]
        pcm, ch = decoded[name][:2]
        rate = decoded[name][2] if len(decoded[name]) > 2 else 44100
        assert (r["file_type_S"], r["file_sample_rate_R"], r["file_channel_count_R"], r["file_bit_depth_R"]) == ("wav", rate, ch, 16)
        assert abs(r["file_length_R"] - pcm.reshape(-1).size / ch / float(rate)) < 1e-6       # SamplesToMs is float arithmetic
        mono, info = _oracle.load_sample(pcm, ch, file_rate=rate)
        ref = ora.run(mono, cap=True)
        for field, base in DB_NAME.items():
            a, b = FIELDS[field]
            rtol, atol = _tol.GPU_TOL[field]
            col = base + ("_VR" if b - a == 1 else "_VVR")
            got = np.array(msgpack.unpackb(r[col]), dtype=np.float64).reshape(ref.shape[0], -1)
            _tol.check_gpu(field, got, ref[:, a:b], rtol, atol, what=f"{name} {col} ")
            # statistics columns: the mean of the series
            want_mean = ref[:, a:b].mean(axis=0)
            if b - a == 1:
                assert abs(r[base + "_mean_R"] - want_mean[0]) <= 1e-4 * abs(want_mean[0]) + 1e-7
            else:
                got_mean = np.array(msgpack.unpackb(r[base + "_mean_VR"]))
                assert np.all(np.abs(got_mean - want_mean) <= 1e-4 * np.abs(want_mean) + 1e-7)
        eff = ora.effective_length(mono)
        assert abs(r["effectve_length_48dB_R"] - eff[0]) < 1e-9
        # the rhythm tracker's columns: onsets and the scalars, with the file's own duration and the
        # data offset LoadSample produced (SampleAnalyser.cpp:1001-1004)
        rh = ora.run_rhythm(mono, original_rate=rate, original_samples=pcm.reshape(-1).size // ch, data_offset=info["data_offset"], cap=True)
        for t, kind in enumerate(("rhythm_complex", "rhythm_percussive")):
            got = np.array(msgpack.unpackb(r[kind + "_onsets_VR"]), dtype=np.float64)
            assert got.shape == rh["onsets"][t].shape
            assert np.array_equal(np.nonzero(got)[0], np.nonzero(rh["onsets"][t])[0]), (name, kind)
            assert np.all(np.abs(got - rh["onsets"][t]) <= 1e-5 * np.abs(rh["onsets"][t]) + 1e-6)
            assert abs(r[kind + "_onsets_mean_R"] - got.mean()) <= 1e-9 * abs(got.mean()) + 1e-12
        sc = dict(zip(_oracle.RHYTHM_SCALARS, rh["scalars"]))
        for key, want in sc.items():
            assert r[key + "_R"] is not None and abs(r[key + "_R"] - want) <= 1e-5 * abs(want) + 1e-9, (name, key, r[key + "_R"], want)
    # the WAV sample types: the normalised buffer's length is what the reference's converters give
    for k in ("u8_mono", "i24_mono", "i32_stereo", "f32_mono", "f64_stereo"):
        r = rows[f"Formats/{k}.wav"]
        assert r["status"] == "succeeded" and r["file_bit_depth_R"] == int(z["props_" + k][2])
        mono = z["data_" + k]
        ref = ora.run(mono, cap=True)
        got = np.array(msgpack.unpackb(r["cepstrum_bands_VVR"]))
        a, b = FIELDS["mfcc"]
        _tol.check_gpu("mfcc", got, ref[:, a:b], *_tol.GPU_TOL["mfcc"], what=k + " ")
    con.close()


def test_crawl_without_a_database_and_small_batches():
    images, names, _ = make_crawl(40)
    a = _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=1000)
    b = _host.crawl(images, names, devices=(0,), workers=4, files_per_batch=3)
    for k in ("files", "failed", "frames", "pcm_bytes", "result_bytes"):
        assert a[k] == b[k], k
    assert a["failed"] == 1 and a["skipped_sample_rate"] == 0 and a["frames"] > 0 and a["writer_seconds"] == 0.0


def test_batches_end_at_the_byte_budget():
    """TCrawlOptions::mBytesPerBatch: a batch ends before the file that would take it over the budget (long files must not
    grow the staging buffer and the device workspace without bound); the results do not depend on it."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(40)
    whole = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=1000)
    assert whole["batches"] == 1
    largest = max(len(b) for b in images)
    try:
        hostlib.set_bytes_per_batch(2 * largest)
        cut = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=1000)
        hostlib.set_bytes_per_batch(1)                 # one file per batch: a batch always takes at least one
        single = _host.crawl(images, names, devices=(0,), workers=3, files_per_batch=1000)
    finally:
        hostlib.set_bytes_per_batch(0)
    assert 5 < cut["batches"] < len(images) and single["batches"] == len(images)
    for k in ("files", "failed", "frames", "pcm_bytes", "result_bytes"):
        assert whole[k] == cut[k] == single[k], k


def test_crawler_persists_between_crawls_and_can_be_released(tmp_path):
    """The process keeps its crawler (plans, device workspaces, page-locked buffers) between crawls: a second crawl and
    a crawl after afec_crawl_release write the same database rows as the first."""
    import hashlib
    images, names, _ = make_crawl(30)

    def rows_of(db):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("PRAGMA table_info(assets)")]
        out = {}
        for r in con.execute("SELECT * FROM assets ORDER BY filename"):
            h = hashlib.sha256()
            for c, v in zip(cols, r):
                if c != "modtime":
                    h.update(repr(v).encode() if not isinstance(v, bytes) else v)
            out[r[0]] = h.hexdigest()
        con.close()
        return out

    dbs = [str(tmp_path / f"crawl{i}.db") for i in range(3)]
    _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=8, database=dbs[0])
    _host.crawl(images, names, devices=(0,), workers=3, files_per_batch=16, database=dbs[1])     # warm crawler, other batching
    from afec_amd import hostlib
    hostlib.release()
    _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=64, database=dbs[2])     # a fresh crawler again
    a, b, c = (rows_of(d) for d in dbs)
    assert len(a) == len(images) and a == b == c


def rows_by_hash(db):
    import hashlib
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(assets)")]
    out = {}
    for r in con.execute("SELECT * FROM assets ORDER BY filename"):
        h = hashlib.sha256()
        for c, v in zip(cols, r):
            if c != "modtime":
                h.update(repr(v).encode() if not isinstance(v, bytes) else v)
        out[r[0]] = h.hexdigest()
    con.close()
    return out


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_several_shards_on_one_device(tmp_path, shards):
    """The G > 1 branch of the crawler (file i -> shard i mod G, one analyser, worker set and cursor per shard,
    Crawler.cpp:706-728) with every shard on device 0: the database rows are those of the one-shard crawl, the files
    go where afec_shard_of_file says, and the crawl reports the host CPUs it kept busy."""
    images, names, _ = make_crawl(90)
    dbs = [str(tmp_path / "one.db"), str(tmp_path / "many.db")]
    one = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=8, database=dbs[0])
    many = _host.crawl(images, names, devices=(0,) * shards, workers=2, files_per_batch=8, database=dbs[1])
    L = _host.lib()
    want = [0] * shards
    for i in range(len(images)):
        want[L.afec_shard_of_file(i, shards)] += 1
    assert many["files_per_device"] == want and sum(want) == len(images)
    for k in ("files", "failed", "frames", "pcm_bytes", "skipped_sample_rate"):
        assert one[k] == many[k], k
    a, b = rows_by_hash(dbs[0]), rows_by_hash(dbs[1])
    assert len(a) == len(images) and a == b
    assert many["cpu_seconds"] > 0 and many["seconds"] > 0


def test_crawl_is_sharded_over_devices():
    import ctypes
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) != 0 or n.value < 2:
        pytest.skip("needs two GPUs")
    images, names, _ = make_crawl(60)
    one = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=8)
    two = _host.crawl(images, names, devices=(0, 1), workers=2, files_per_batch=8)
    assert two["files_per_device"] == [(len(images) + 1) // 2, len(images) // 2]
    for k in ("files", "failed", "frames"):
        assert one[k] == two[k]


def test_database_pragmas_change_how_sqlite_writes_not_what(tmp_path):
    """TCrawlOptions::mDatabasePragmas (larger pages, journal in memory, no fsync per commit: about twice the writer's
    rate): the same rows, a plain sqlite file."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(40)
    dbs = [str(tmp_path / "default.db"), str(tmp_path / "tuned.db")]
    _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=16, database=dbs[0])
    try:
        hostlib.set_database_pragmas("PRAGMA page_size=65536; PRAGMA journal_mode=MEMORY; PRAGMA synchronous=OFF")
        _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=16, database=dbs[1])
    finally:
        hostlib.set_database_pragmas("")
    a, b = rows_by_hash(dbs[0]), rows_by_hash(dbs[1])
    assert len(a) == len(images) and a == b
    con = sqlite3.connect(dbs[1])
    assert con.execute("PRAGMA page_size").fetchone()[0] == 65536 and con.execute("PRAGMA user_version").fetchone()[0] == 2
    con.close()


def test_files_on_disk_give_the_rows_of_their_images(tmp_path):
    """The crawler reads files on disk itself (TCrawlFile::mpImage == nullptr: head parsed, data chunk pread into the
    page-locked staging buffer): the same rows as for the same bytes handed over as images under the same names; a
    path that does not exist becomes a failed sample with the reference's text (RiffFile.cpp:128-131)."""
    images, names, _ = make_crawl(60)
    paths = []
    for name, image in zip(names, images):
        p = tmp_path / "files" / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(image)
        paths.append(str(p))
    dbs = [str(tmp_path / "images.db"), str(tmp_path / "disk.db")]
    from_images = _host.crawl(images, paths, workers=2, files_per_batch=16, database=dbs[0])
    from_disk = _host.crawl(None, paths + [str(tmp_path / "files" / "gone.wav")], workers=2, files_per_batch=16, database=dbs[1])
    assert from_disk["files"] == from_images["files"] + 1 and from_disk["failed"] == from_images["failed"] + 1
    for k in ("frames", "pcm_bytes", "skipped_sample_rate"):
        assert from_disk[k] == from_images[k], k
    a, b = rows_by_hash(dbs[0]), rows_by_hash(dbs[1])
    gone = b.pop(str(tmp_path / "files" / "gone.wav"))
    assert gone and a == b and len(a) == len(images)
    con = sqlite3.connect(dbs[1])
    status = con.execute("SELECT status FROM assets WHERE filename = ?", (str(tmp_path / "files" / "gone.wav"),)).fetchone()[0]
    con.close()
    assert "Failed to open the file" in status


def test_the_c4_share_at_its_full_size(tmp_path):
    """BASELINE.json configs[3]'s per-GPU share at its real size: 12 500 stereo one-second files (64 different contents,
    cycled) through the crawler into the database.  Size-independent properties: every file is there, files with the
    same bytes have the same row whatever batch, worker and position they were analysed in, and a sample of the
    contents equals the oracle."""
    import hashlib
    import msgpack
    import bench
    contents = bench.make_c4_files(64, 99)
    pool = [bench.wav_image(f, 2) for f in contents]
    n = 12500
    images = [pool[i % 64] for i in range(n)]
    names = [f"share/{i // 500:02d}/file{i:05d}.wav" for i in range(n)]
    db = str(tmp_path / "c4.db")
    st = _host.crawl(images, names, workers=8, files_per_batch=512, database=db)
    assert st["files"] == n and st["failed"] == 0 and st["batches"] >= n // 512
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(assets)")]
    skip = {cols.index("filename"), cols.index("modtime")}
    digest, count = {}, 0
    for r in con.execute("SELECT * FROM assets"):
        count += 1
        h = hashlib.sha256()
        for k, v in enumerate(r):
            if k not in skip:
                h.update(v if isinstance(v, bytes) else repr(v).encode())
        i = int(r[0][-9:-4])
        digest.setdefault(i % 64, set()).add(h.hexdigest())
    assert count == n and len(digest) == 64
    assert all(len(v) == 1 for v in digest.values()), [k for k, v in digest.items() if len(v) != 1]
    con.row_factory = sqlite3.Row
    ora = Oracle()
    for k in (0, 17, 63):
        r = con.execute("SELECT * FROM assets WHERE filename = ?", (names[64 * 100 + k],)).fetchone()
        mono, _ = _oracle.load_sample(contents[k], 2)
        ref = ora.run(mono, cap=True)
        a, b = FIELDS["mfcc"]
        got = np.array(msgpack.unpackb(r["cepstrum_bands_VVR"]), dtype=np.float64).reshape(ref.shape[0], -1)
        _tol.check_gpu("mfcc", got, ref[:, a:b], *_tol.GPU_TOL["mfcc"], what=f"content {k} ")
        assert r["status"] == "succeeded" and r["file_channel_count_R"] == 2
    con.close()


def test_conversion_can_be_switched_off():
    """TCrawlOptions::mResample = false: files at another rate are skipped and counted, not analysed, not failed."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(20)
    on = _host.crawl(images, names, workers=2, files_per_batch=8)
    try:
        hostlib.set_resample(False)
        off = _host.crawl(images, names, workers=2, files_per_batch=8)
    finally:
        hostlib.set_resample(True)
    assert on["skipped_sample_rate"] == 0 and off["skipped_sample_rate"] == 3
    assert off["files"] == on["files"] and off["failed"] == on["failed"] == 1 and off["frames"] < on["frames"]


def statuses(db):
    con = sqlite3.connect(db)
    out = {r[0]: r[1] for r in con.execute("SELECT filename, status FROM assets")}
    con.close()
    return out


def test_a_failed_gpu_round_trip_is_retried_in_halves_and_the_crawl_goes_on(tmp_path):
    """SampleAnalyser.cpp:368-408: a file that cannot be analysed gets a failed row and the crawl continues.  For errors
    of the device path (out of memory, a failed runtime call) the unit that fails is a batch: it is retried in halves
    (TCrawlOptions::mTestFailBatch injects the fault).  One failure: every file still gets its row, identical to the
    clean crawl's.  A batch that fails whatever its size: its files become "Sample failed to analyse" rows, every other
    row is the clean crawl's, the crawl ends normally."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(90)        # 90 + 5 + 1 + 3 files, batches of 16: batch 3 = files 48..63
    dbs = [str(tmp_path / f"{k}.db") for k in ("clean", "once", "always")]
    clean = _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=16, database=dbs[0])
    assert clean["retried_batches"] == 0 and clean["device_failed_files"] == 0
    try:
        hostlib.set_test_fault(batch=3, attempts=1)
        once = _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=16, database=dbs[1])
        hostlib.set_test_fault(batch=3, attempts=-1)
        always = _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=16, database=dbs[2])
    finally:
        hostlib.set_test_fault()
    a, b, c = (rows_by_hash(d) for d in dbs)
    assert len(a) == len(images) and a == b
    assert once["retried_batches"] == 1 and once["device_failed_files"] == 0 and once["failed"] == clean["failed"]
    assert once["frames"] == clean["frames"] and once["batches"] == clean["batches"] + 1        # two halves instead of one batch
    hit = set(names[48:64])
    assert len(c) == len(images) and {n: h for n, h in c.items() if n not in hit} == {n: h for n, h in a.items() if n not in hit}
    st = statuses(dbs[2])
    assert all(st[n].startswith("error: Sample failed to analyse: ") and "injected fault" in st[n] for n in hit)
    assert always["device_failed_files"] == 16 and always["failed"] == clean["failed"] + 16
    # 1 whole + 2 + 4 + 8 halves + 16 single files tried twice = 47 failed round trips
    assert always["retried_batches"] == 47


def test_a_lost_device_ends_the_crawl():
    """...whereas a device that no longer answers (TCrawlOptions::mTestDeviceLost stands in for the probe's verdict) is
    not something a smaller batch cures: the crawl ends with the error."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(40)
    try:
        hostlib.set_test_fault(batch=1, attempts=1, device_lost=True)
        with pytest.raises(RuntimeError, match="injected fault"):
            _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=8)
    finally:
        hostlib.set_test_fault()
    again = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=8)     # the crawler itself is fine
    assert again["files"] == len(images) and again["failed"] == 1


def test_a_crawl_that_is_ending_commits_whole_batches_only(tmp_path):
    """A worker that loses its device sets the crawl's abort flag while the writer may be in the middle of another batch:
    until round 5 the writer's loop then left early and the commit behind it committed the partial batch.  With batches of
    8 files in file order (one worker): whatever reached the database before the crawl ended is whole batches -- every
    group of 8 consecutive files is either complete or absent."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(120)
    db = str(tmp_path / "ending.db")
    try:
        hostlib.set_test_fault(batch=9, attempts=1, device_lost=True)
        with pytest.raises(RuntimeError, match="injected fault"):
            _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=8, database=db)
    finally:
        hostlib.set_test_fault()
    have = set(statuses(db))
    groups = [names[i:i + 8] for i in range(0, len(names), 8)]
    counts = [sum(n in have for n in g) for g in groups]
    assert all(c in (0, len(g)) for c, g in zip(counts, groups)), counts
    assert 0 < sum(counts) < len(names)                     # some batches made it, the crawl did end early


def test_a_file_whose_header_would_blow_it_up_fails_alone(tmp_path):
    """A small file whose header claims 1 Hz would be 2^31 samples once converted to 44.1 kHz: it is refused by itself
    (AFX_ERR_UNSUPPORTED -> a failed row), its batch is analysed; files at 1 kHz (x 44.1) and 500 Hz (x 88.2: refused
    until round 5, when any rate below the analyser's / 64 was) are converted like the reference converts them, and the
    batch they are in is cut by the device budget (TCrawlOptions::mDeviceBytesPerBatch) instead of growing the workspace."""
    from afec_amd import hostlib
    images, names, _ = make_crawl(20)
    rng = np.random.default_rng(4)
    tiny = np.round(rng.uniform(-0.5, 0.5, 48000) * 32767).astype(np.int16)
    images += [wav_bytes(tiny, 1, 16, rate=1), wav_bytes(tiny[:4000], 1, 16, rate=1000), wav_bytes(tiny[:4000], 1, 16, rate=500)]
    names += ["Odd/one_hertz.wav", "Odd/one_kilohertz.wav", "Odd/five_hundred_hertz.wav"]
    db = str(tmp_path / "odd.db")
    st = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=64, database=db)
    s = statuses(db)
    assert s["Odd/one_hertz.wav"].startswith("error: Sample failed to load: ")
    assert s["Odd/one_kilohertz.wav"] == "succeeded" and s["Odd/five_hundred_hertz.wav"] == "succeeded"
    assert st["files"] == len(images) and st["failed"] == 2 and st["retried_batches"] == 0
    try:
        hostlib.set_device_bytes_per_batch(4 << 20)          # the 1 kHz file alone is 176 400 converted samples = 2.8 MB
        cut = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=64)
    finally:
        hostlib.set_device_bytes_per_batch(0)
    assert cut["batches"] > st["batches"] and cut["frames"] == st["frames"] and cut["failed"] == 2



def test_a_files_row_does_not_depend_on_the_batch_it_was_crawled_in(tmp_path):
    """The library's AUTO picks the STFT kernel's layout by batch size (~32 768 frames) and the two layouts round
    differently: with AUTO in the crawler a file's database row depended on the batch it landed in -- the tail batch of a
    crawl, the halves of a retried batch.  The crawler pins the layout (TCrawlOptions::mFrameKernel): 600 two-second
    files crawled in batches of 512 (43 000 frames: above the threshold, + a tail batch of 88 files below it) and in
    batches of 64 give byte-identical rows; with AUTO forced the rows differ, which is what the pin is for."""
    from afec_amd import hostlib
    rng = np.random.default_rng(93)
    pool = []
    for _ in range(8):
        x, ch = synth(rng, 2.0, stereo=False)
        pool.append(wav_bytes(np.round(x * 32767).astype(np.int16), ch, 16))
    images = [pool[i % 8] for i in range(600)]
    names = [f"pin/file{i:04d}.wav" for i in range(600)]
    dbs = [str(tmp_path / f"{k}.db") for k in ("big", "small", "auto_big", "auto_small")]
    _host.crawl(images, names, workers=2, files_per_batch=512, database=dbs[0])
    _host.crawl(images, names, workers=2, files_per_batch=64, database=dbs[1])
    a, b = rows_by_hash(dbs[0]), rows_by_hash(dbs[1])
    assert len(a) == 600 and a == b
    # the same 8 contents: every copy of a content has one and the same row
    con = sqlite3.connect(dbs[0])
    blobs = {}
    for name, blob in con.execute("SELECT filename, cepstrum_bands_VVR FROM assets"):
        blobs.setdefault(int(name[-8:-4]) % 8, set()).add(blob)
    con.close()
    assert all(len(v) == 1 for v in blobs.values())
    try:
        hostlib.set_frame_kernel(0)   # AFX_FRAME_KERNEL_AUTO: what the crawler ran until round 5
        _host.crawl(images, names, workers=2, files_per_batch=512, database=dbs[2])
        _host.crawl(images, names, workers=2, files_per_batch=64, database=dbs[3])
    finally:
        hostlib.set_frame_kernel(-1)
    c, d = rows_by_hash(dbs[2]), rows_by_hash(dbs[3])
    assert d == b                    # small batches: the 64-lane layout either way
    assert c != d                    # ... and AUTO's large batches took the other layout: rows that depend on the batch


def test_an_external_abort_ends_the_crawl_early_and_leaves_whole_batches(tmp_path):
    """The reference's SIGINT flag (Crawler.cpp:69-73, 717-720: every task checks sAbortProcessing before it starts):
    afec_crawl_request_abort from another thread while a crawl with the database on is running -- workers take no new
    batch, what was analysed is written (whole batches only), the call returns normally with `aborted` set.  The same
    scenario runs under ThreadSanitizer on the mock device (tests/sanitize/tsan_crawler.cpp)."""
    import threading
    from afec_amd import hostlib
    images, names, _ = make_crawl(120)
    images, names = images * 12, [f"{k:02d}/{n}" for k in range(12) for n in names]      # ~1 550 files, batches of 8
    db = str(tmp_path / "aborted.db")
    _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=8, database=str(tmp_path / "cold.db"))   # sets the crawler up
    clean = _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=8, database=str(tmp_path / "clean.db"))
    assert not clean["aborted"] and clean["files"] == len(images)
    st = None
    for attempt, part in enumerate((3, 6, 12, 24)):      # (the request must land while the crawl runs: earlier if it did not)
        db = str(tmp_path / f"aborted{attempt}.db")
        timer = threading.Timer(clean["seconds"] / part, hostlib.request_abort)
        timer.start()
        st = _host.crawl(images, names, devices=(0,), workers=1, files_per_batch=8, database=db)
        timer.join()
        if st["aborted"] and 0 < st["files"] < len(images):
            break
    assert st["aborted"] and 0 < st["files"] < len(images), st
    have = set(statuses(db))
    assert len(have) == st["files"]
    groups = [names[i:i + 8] for i in range(0, len(names), 8)]
    counts = [sum(n in have for n in g) for g in groups]
    assert all(c in (0, len(g)) for c, g in zip(counts, groups)), counts
    # the next crawl is not affected by the old request
    again = _host.crawl(images[:40], names[:40], devices=(0,), workers=2, files_per_batch=8)
    assert not again["aborted"] and again["files"] == 40


def test_row_digests_are_a_function_of_the_content_not_of_the_shard_or_the_batch():
    """TCrawlOptions::mRowDigests (bench.py's sharded crawl checks its devices with it): a 64-bit digest per file over
    everything the device returned.  The same 63 contents crawled as two shards of device 0 in batches of 16, as one shard
    in batches of 50 and with another worker count: a content's digest is the same everywhere, different contents
    differ, a file that cannot be read has none."""
    assert afx.device_count() >= 1
    images, names, _ = make_crawl(62)
    images, names = images[:63], names[:63]
    reps = 4
    many, many_names = images * reps, [f"{k}/{n}" for k in range(reps) for n in names]
    a = _host.crawl(many, many_names, devices=(0, 0), workers=2, files_per_batch=16, digests=True)
    b = _host.crawl(many, many_names, devices=(0,), workers=3, files_per_batch=50, digests=True)
    assert a["files_per_device"] == [126, 126] and b["files_per_device"] == [252]
    da, db = a["row_digests"].reshape(reps, 63), b["row_digests"].reshape(reps, 63)
    assert np.array_equal(da, np.broadcast_to(da[0], da.shape)) and np.array_equal(db, da)
    analysed = da[0][da[0] != 0]
    assert analysed.size >= 60 and np.unique(analysed).size == analysed.size
