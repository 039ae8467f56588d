"""Real audio: thirteen of the reference's own test fixtures (tests/golden/wav/, from
Source/Crawler/XUnitTests/Resources/Kicks-vs-Snare-{Train,Test}, the folders UnitTests.cpp:152-423 crawls): drum
one-shots, 16- and 24-bit, mono and stereo, extra RIFF chunks, the near-silent "_empty wave file.wav", a file name
outside ASCII and "_Not A Wavefile.wav".  Expected values (tests/golden/real.npz, tests/golden/make_golden_real.py)
come from the reference's own objects: LoadSample's converters, LibXtract / Ooura / TStatistics per frame, aubio /
TEnvelopeDetector / TAutocorrelation neighbours, the onset STFT front end and aubio's beat tracker.

CPU: the host WAV reader and the oracle chain against those goldens.  GPU: the files crawled through
afec::CrawlWaveFiles into the descriptor database, EVERY low-level column of every row compared -- per-frame series
with the reference-generated goldens, per-file statistics with TStatistics::Calc's restatement on the golden series,
rhythm columns with the oracle (its detector part is parity-unpinned)."""
import glob
import os
import sqlite3

import numpy as np
import pytest

from tests import _host, _oracle, _tol
from tests._oracle import FIELDS, NEIGH_FIELDS, Oracle
from tests._wav import parse_wav

HERE = os.path.dirname(__file__)
WAV_DIR = os.path.join(HERE, "golden", "wav")
GOLD = os.path.join(HERE, "golden", "real.npz")
ORACLE_RTOL = 1e-6     # oracle vs reference objects: another FFT algorithm

# C-ABI / oracle series name -> descriptor base name in the database (SampleDescriptors.cpp:150-205)
SPECTRAL_DB = {"mfcc": "cepstrum_bands", "spectral_rms": "spectral_rms", "spectral_centroid": "spectral_centroid",
               "spectral_spread": "spectral_spread", "spectral_skewness": "spectral_skewness",
               "spectral_kurtosis": "spectral_kurtosis", "spectral_rolloff": "spectral_rolloff",
               "spectral_flatness": "spectral_flatness", "spectral_flux": "spectral_flux", "spectrum_bands": "frequency_bands",
               "sub_rms": "spectral_rms_bands", "sub_flatness": "spectral_flatness_bands", "sub_flux": "spectral_flux_bands",
               "sub_complexity": "spectral_complexity_bands", "sub_contrast": "spectral_contrast_bands",
               "spectral_contrast": "spectral_contrast", "amplitude_peak": "amplitude_peak", "amplitude_rms": "amplitude_rms"}
STATS = ["min", "max", "median", "mean", "gmean", "variance", "centroid", "spread", "skewness", "kurtosis", "flatness",
         "dmean", "dvariance"]


def files():
    z = np.load(GOLD)
    return [(i, str(n)) for i, n in enumerate(z["names"])]


def decoded(name):
    channels, rate, bits, frames, payload = parse_wav(open(os.path.join(WAV_DIR, name), "rb").read())
    data = np.frombuffer(payload, dtype=np.int16 if bits == 16 else np.uint8)
    return data, channels, frames


def test_the_fixture_set_is_what_the_generator_saw():
    on_disk = sorted(os.path.basename(p) for p in glob.glob(os.path.join(WAV_DIR, "*.wav")))
    assert on_disk == sorted([n for _, n in files()] + ["_Not A Wavefile.wav"])


@pytest.mark.parametrize("index,name", files())
def test_host_reader_on_the_reference_fixtures(index, name):
    z = np.load(GOLD)
    image = open(os.path.join(WAV_DIR, name), "rb").read()
    props, payload = _host.wave_probe(image)
    channels, rate, bits, frames = [int(v) for v in z[f"props_{index}"]]
    assert (props["channels"], props["rate"], props["bits"], props["frames"]) == (channels, rate, bits, frames)
    assert payload == parse_wav(image)[4]


def test_host_reader_rejects_the_broken_fixture():
    with pytest.raises(RuntimeError) as ei:
        _host.wave_probe(open(os.path.join(WAV_DIR, "_Not A Wavefile.wav"), "rb").read())
    assert str(ei.value) == "Not a valid WAV file."           # UnitTests.cpp:338-350: the one failed sample of the set


@pytest.mark.parametrize("index,name", files())
def test_oracle_chain_on_the_reference_fixtures(index, name):
    z = np.load(GOLD)
    data, channels, frames = decoded(name)
    mono, info = _oracle.load_sample(data, channels)
    assert [info["data_offset"], info["silent_leading"], info["silent_trailing"], info["n_samples"]] == z[f"info_{index}"].tolist()
    pr = z[f"peakrms_{index}"]
    assert info["peak_value"] == pr[0] and abs(info["rms_value"] - pr[1]) <= 1e-6 * pr[1]
    ora = Oracle()
    rec = ora.run(mono, cap=True)
    want = z[f"spectral_{index}"]
    assert rec.shape[0] == want.shape[0]
    for field, (a, b) in FIELDS.items():
        if field == "mag":
            continue
        rtol, atol = _tol.GPU_TOL[field]
        _tol.check(field, rec[:, a:b], want[:, a - 1024:b - 1024], min(rtol, ORACLE_RTOL) if rtol else 0.0, atol, what=f"{name} oracle ")
    nei = ora.run_neighbours(mono, cap=True)
    for field, col in NEIGH_FIELDS.items():
        rtol, atol = _tol.NEIGH_TOL[field]
        _tol.check(field, nei[:, col], z[f"neighbours_{index}"][:, col], rtol, atol, what=f"{name} oracle ")
    # the rhythm tracker's STFT front end on the same buffer
    sel, ref = z[f"polar_frames_{index}"], z[f"polar_{index}"]
    got = np.stack([_oracle.onset_polar(mono[f * 128:f * 128 + 512]) for f in sel])
    mag_ref, mag_got = ref[:, :257].astype(np.float64), got[:, :257].astype(np.float64)
    scale = np.abs(mag_ref[:, 2:]).max(axis=1, keepdims=True) + 1e-30
    assert np.all(np.abs(mag_ref - mag_got) <= 1.2e-7 * np.abs(mag_ref) + 1e-13 * scale)
    loud = mag_ref[:, 2:] > 1e-9 * scale
    d = np.abs(ref[:, 257:].astype(np.float64) - got[:, 257:].astype(np.float64))
    d = np.minimum(d, 2 * np.pi - d)
    assert np.all(d[loud] <= 5e-7)


def test_beat_tracking_of_the_fixtures_matches_reference_aubio():
    z = np.load(GOLD)
    for key, ref in zip(z["beat_keys"], z["beat_out"]):
        bpm, conf = _oracle.beattrack(z[f"beat_in_{key}"])
        assert bpm == pytest.approx(ref[0], rel=1e-13, abs=0), key
        assert conf == pytest.approx(ref[1], rel=1e-12, abs=1e-300), key


def stat_tolerance(series, want):
    """Statistics of a GPU-produced series against TStatistics::Calc's restatement on the reference's series: the
    series themselves agree to 1e-4 relative (the north-star bar), so a statistic agrees to that times its
    condition number; the third / fourth moments of a short series amplify most."""
    scale = max(1e-300, float(np.max(np.abs(series))) if series.size else 0.0)
    tol = np.full(13, 1e-3)
    floor = np.full(13, 1e-4 * scale + 1e-9)
    floor[5] = floor[12] = 1e-4 * scale * scale + 1e-12        # variance, dvariance
    floor[6] = floor[7] = 1e-2 * max(1, series.size) ** 2      # centroid / spread: in frames
    floor[8] = floor[9] = np.inf                                # skewness / kurtosis: see the separate check
    return tol * np.abs(want) + floor


@pytest.mark.gpu
def test_crawl_the_reference_fixtures_into_the_database(tmp_path):
    import msgpack
    z = np.load(GOLD)
    names = ["Kicks/" + n for _, n in files()] + ["Kicks/_Not A Wavefile.wav"]
    images = [open(os.path.join(WAV_DIR, os.path.basename(n)), "rb").read() for n in names]
    db = str(tmp_path / "real.db")
    st = _host.crawl(images, names, devices=(0,), workers=2, files_per_batch=5, database=db)
    assert st["files"] == len(images) and st["failed"] == 1 and st["skipped_sample_rate"] == 0
    con = sqlite3.connect(db)
    con.row_factory = sqlite3.Row
    rows = {r["filename"]: r for r in con.execute("SELECT * FROM assets")}
    assert len(rows) == len(images)
    assert rows["Kicks/_Not A Wavefile.wav"]["status"] == "error: Sample failed to load: Not a valid WAV file."
    assert sum(1 for r in rows.values() if r["status"] != "succeeded") == 1            # UnitTests.cpp:338-350
    ora = Oracle()
    checked = set()
    for index, base in files():
        r = rows["Kicks/" + base]
        checked.update({"filename", "modtime", "status"})
        channels, rate, bits, frames = [int(v) for v in z[f"props_{index}"]]
        assert (r["file_type_S"], r["file_sample_rate_R"], r["file_channel_count_R"], r["file_bit_depth_R"]) == ("wav", rate, channels, bits)
        assert r["file_size_R"] == len(images[index]) and abs(r["file_length_R"] - frames / rate) < 1e-12
        checked.update({"file_type_S", "file_sample_rate_R", "file_channel_count_R", "file_bit_depth_R", "file_size_R", "file_length_R"})
        info = z[f"info_{index}"]
        # analyzation_offset: TAudioMath::SamplesToMs in float / 1000 (SampleAnalyser.cpp:748-749)
        want_off = float(np.float32(info[0]) / (np.float32(44100) / np.float32(1000.0))) / 1000.0
        assert abs(r["analyzation_offset_R"] - want_off) <= 1e-12
        checked.add("analyzation_offset_R")
        want, nwant = z[f"spectral_{index}"], z[f"neighbours_{index}"]
        F = want.shape[0]

        def series_and_stats(db_base, field, got_ref, rtol, atol, width):
            col = db_base + ("_VR" if width == 1 else "_VVR")
            got = np.array(msgpack.unpackb(r[col]), dtype=np.float64).reshape(F, -1)
            _tol.check_gpu(field, got, got_ref, rtol, atol, what=f"{base} {col} ")
            checked.add(col)
            for w in range(width):
                ref13 = _oracle.calc_statistics(got_ref[:, w])
                tol13 = stat_tolerance(got_ref[:, w], ref13)
                own13 = _oracle.calc_statistics(got[:, w])            # Calc of the GPU's own series: the kernel itself
                for k, sname in enumerate(STATS):
                    if width == 1:
                        v = r[f"{db_base}_{sname}_R"]
                    else:
                        v = msgpack.unpackb(r[f"{db_base}_{sname}_VR"])[w]
                    checked.add(f"{db_base}_{sname}_" + ("R" if width == 1 else "VR"))
                    assert np.isfinite(v), (base, db_base, sname)
                    assert abs(v - ref13[k]) <= tol13[k], (base, db_base, sname, w, v, ref13[k])
                    if k not in (8, 9):
                        assert abs(v - own13[k]) <= 1e-8 * abs(own13[k]) + 1e-9 * (1.0 + np.max(np.abs(got[:, w]))), (base, db_base, sname, w, v, own13[k])
                    else:
                        # skewness / kurtosis divide by sigma^3 / sigma^4: compared where the series is not near-constant
                        spread = np.ptp(got[:, w])
                        if spread > 1e-6 * (1e-300 + np.max(np.abs(got[:, w]))):
                            assert abs(v - own13[k]) <= 1e-6 * abs(own13[k]) + 1e-6, (base, db_base, sname, w, v, own13[k])

        for field, db_base in SPECTRAL_DB.items():
            a, b = FIELDS[field]
            rtol, atol = _tol.GPU_TOL[field]
            series_and_stats(db_base, field, want[:, a - 1024:b - 1024], rtol, atol, b - a)
        for field, col in NEIGH_FIELDS.items():
            rtol, atol = _tol.NEIGH_TOL[field]
            series_and_stats(field, field, nwant[:, col:col + 1], rtol, atol, 1)
        # effective lengths and the rhythm tracker: against the oracle on the reference-loaded buffer
        data, ch, nframes = decoded(base)
        mono, oinfo = _oracle.load_sample(data, ch)
        eff = ora.effective_length(mono)
        for k, col in enumerate(("effectve_length_48dB_R", "effectve_length_24dB_R", "effectve_length_12dB_R")):
            assert abs(r[col] - eff[k]) < 1e-9, (base, col)
            checked.add(col)
        rh = ora.run_rhythm(mono, original_samples=nframes, data_offset=oinfo["data_offset"], cap=True)
        for t, kind in enumerate(("rhythm_complex", "rhythm_percussive")):
            got = np.array(msgpack.unpackb(r[kind + "_onsets_VR"]), dtype=np.float64)
            assert got.shape == rh["onsets"][t].shape
            assert np.array_equal(np.nonzero(got)[0], np.nonzero(rh["onsets"][t])[0]), (base, kind)
            assert np.all(np.abs(got - rh["onsets"][t]) <= 1e-5 * np.abs(rh["onsets"][t]) + 1e-6)
            checked.add(kind + "_onsets_VR")
            own13 = _oracle.calc_statistics(got)
            for k, sname in enumerate(STATS):
                v = r[f"{kind}_onsets_{sname}_R"]
                checked.add(f"{kind}_onsets_{sname}_R")
                if k in (8, 9) and np.ptp(got) <= 1e-6 * (1e-300 + np.max(np.abs(got))):
                    continue
                assert abs(v - own13[k]) <= 1e-6 * abs(own13[k]) + 1e-9 * (1.0 + np.max(np.abs(got))), (base, kind, sname, v, own13[k])
        sc = dict(zip(_oracle.RHYTHM_SCALARS, rh["scalars"]))
        for key, w in sc.items():
            assert r[key + "_R"] is not None and abs(r[key + "_R"] - w) <= 1e-5 * abs(w) + 1e-9, (base, key, r[key + "_R"], w)
            checked.add(key + "_R")
    # every column of the table was compared for every succeeded row
    all_columns = [c[1] for c in con.execute("PRAGMA table_info(assets)")]
    assert len(all_columns) == 461 and sorted(checked) == sorted(all_columns), sorted(set(all_columns) - checked)
    con.close()
