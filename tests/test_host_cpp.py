"""The C++ host layer (afec_amd/host: TSampleAnalyser / TSampleDescriptors mirror; it computes nothing itself).
The test program links the oracle as the checker; the host library itself only links libafx_hip."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "host_test")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "afec_amd", "csrc")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "afec_amd", "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libafx_oracle.so"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "afec_amd", "lib")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-o", BIN, os.path.join(ROOT, "tests", "host", "test_sample_analyser.cpp"),
        "-L" + lib, "-lafx_host", "-lafx_hip", "-L" + os.path.join(ROOT, "oracle"), "-lafx_oracle", "-lm",
        "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])


def run(mode):
    build()
    out = subprocess.run([BIN, mode], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_host_layer_fails_loudly_without_a_device():
    run("nodevice")


STATS = ["min", "max", "median", "mean", "gmean", "variance", "centroid", "spread", "skewness", "kurtosis", "flatness",
         "dmean", "dvariance"]
# the reference's low-level columns this library produces, in its column order (README.md "Low-Level Features",
# Source/Crawler/FeatureExtraction/Source/SampleDescriptors.cpp:150-205)
SCALAR_SERIES = ["amplitude_silence", "amplitude_peak", "amplitude_rms", "amplitude_envelope", "spectral_rms",
                 "spectral_centroid", "spectral_rolloff", "spectral_spread", "spectral_skewness", "spectral_kurtosis",
                 "spectral_flatness", "spectral_inharmonicity", "spectral_complexity", "spectral_contrast",
                 "spectral_flux", "f0", "f0_confidence", "failsafe_f0", "tristimulus1", "tristimulus2", "tristimulus3",
                 "auto_correlation"]
VECTOR_SERIES = ["spectral_rms_bands", "spectral_flatness_bands", "spectral_flux_bands", "spectral_complexity_bands",
                 "spectral_contrast_bands", "frequency_bands", "cepstrum_bands"]


def expected_columns():
    names = ["effectve_length_48dB_R", "effectve_length_24dB_R", "effectve_length_12dB_R", "analyzation_offset_R"]
    for n in SCALAR_SERIES:
        names += [n + "_VR"] + [f"{n}_{s}_R" for s in STATS]
    for kind in ("rhythm_complex", "rhythm_percussive"):      # SampleDescriptors.cpp:180-193
        names += [kind + "_onsets_VR"] + [f"{kind}_onsets_{s}_R" for s in STATS]
        names += [kind + s + "_R" for s in ("_onset_count", "_onset_contrast", "_onset_frequency_mean", "_onset_strength",
                                            "_tempo", "_tempo_confidence")]
    names += ["rhythm_final_tempo_R", "rhythm_final_tempo_confidence_R"]
    for n in VECTOR_SERIES:
        names += [n + "_VVR"] + [f"{n}_{s}_VR" for s in STATS]
    return names


def test_column_names_and_msgpack_blobs_match_the_reference(tmp_path):
    """SURVEY 8f/f2, data-format half: the encoder's BLOBs are byte-identical to the reference's vendored msgpack-c
    driven as SToMsgpack drives it (tests/golden/columns.npz), every BLOB decodes with an independent msgpack
    implementation, and the column names / order are the reference's."""
    import hashlib
    import struct

    import msgpack
    import numpy as np
    from tests.golden.make_golden import column_values

    build()
    path = str(tmp_path / "columns.bin")
    out = subprocess.run([BIN, "columns", path], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(path, "rb").read()
    pos = 0

    def take_blob():
        nonlocal pos
        n = struct.unpack_from("<Q", raw, pos)[0]
        pos += 8
        b = raw[pos:pos + n]
        pos += n
        return b

    gold = np.load(os.path.join(ROOT, "tests", "golden", "columns.npz"))
    for n in (0, 1, 15, 16, 860):
        b = take_blob()
        assert b == gold[f"vr_{n}"].tobytes(), n
        assert msgpack.unpackb(b) == column_values(n, 0, n).tolist()
    big = take_blob()
    assert hashlib.sha256(big).digest() == gold["vr_70000_sha256"].tobytes()
    assert big[:14] == gold["vr_70000_head"].tobytes() and big[0] == 0xdd      # array 32
    for rows, width in ((0, 14), (3, 14), (20, 28), (860, 14)):
        b = take_blob()
        assert b == gold[f"vvr_{rows}x{width}"].tobytes(), (rows, width)
        assert msgpack.unpackb(b) == column_values(rows, width, rows + width).tolist()

    count = struct.unpack_from("<Q", raw, pos)[0]
    pos += 8
    cols = {}
    order = []
    for _ in range(count):
        n = struct.unpack_from("<Q", raw, pos)[0]
        pos += 8
        name = raw[pos:pos + n].decode()
        pos += n
        kind = raw[pos]
        pos += 1
        if kind == 0:
            cols[name] = struct.unpack_from("<d", raw, pos)[0]
            pos += 8
        else:
            cols[name] = msgpack.unpackb(take_blob())
        order.append(name)
    assert pos == len(raw)
    assert order == expected_columns()
    assert len(order) == 4 + 22 * 14 + 2 * (14 + 6) + 2 + 7 * 14        # every low-level column but the six file_* ones
    assert cols["rhythm_complex_onsets_VR"] == column_values(7, 0, 4).tolist() and cols["rhythm_final_tempo_R"] == 123.5
    assert cols["effectve_length_48dB_R"] == 1.5 and cols["effectve_length_12dB_R"] == 0.5
    # SamplesToMs in float, then / 1000 (SampleAnalyser.cpp:748-749)
    assert cols["analyzation_offset_R"] == float(np.float32(-2205) / (np.float32(44100) / np.float32(1000))) / 1000.0
    assert cols["spectral_centroid_VR"] == column_values(5, 0, 1).tolist()
    assert cols["spectral_centroid_median_R"] == 2.5 and cols["spectral_centroid_dvariance_R"] == -0.125
    assert cols["f0_VR"] == column_values(5, 0, 2).tolist()
    assert cols["cepstrum_bands_VVR"] == column_values(5, 14, 3).tolist()
    assert cols["cepstrum_bands_mean_VR"] == [b / 4.0 for b in range(14)]
    assert cols["amplitude_peak_VR"] == [] and cols["frequency_bands_VVR"] == []


def full_schema():
    """every column of the reference's low-level `assets` table with its declared type (README.md "Low-Level
    Features"; SampleDescriptors.cpp:150-205; SqliteSampleDescriptorPool.cpp:1313-1358)"""
    cols = [("filename", "TEXT"), ("modtime", "INTEGER"), ("status", "TEXT"), ("file_type_S", "TEXT"),
            ("file_size_R", "INTEGER"), ("file_length_R", "REAL"), ("file_sample_rate_R", "INTEGER"),
            ("file_channel_count_R", "INTEGER"), ("file_bit_depth_R", "INTEGER"), ("effectve_length_48dB_R", "REAL"),
            ("effectve_length_24dB_R", "REAL"), ("effectve_length_12dB_R", "REAL"), ("analyzation_offset_R", "REAL")]

    def framed_scalar(n):
        return [(n + "_VR", "BLOB")] + [(f"{n}_{s}_R", "REAL") for s in STATS]

    for n in SCALAR_SERIES:
        cols += framed_scalar(n)
    for kind in ("rhythm_complex", "rhythm_percussive"):
        cols += framed_scalar(kind + "_onsets")
        cols += [(kind + s + "_R", "REAL") for s in ("_onset_count", "_onset_contrast", "_onset_frequency_mean",
                                                      "_onset_strength", "_tempo", "_tempo_confidence")]
    cols += [("rhythm_final_tempo_R", "REAL"), ("rhythm_final_tempo_confidence_R", "REAL")]
    for n in VECTOR_SERIES:
        cols += [(n + "_VVR", "BLOB")] + [(f"{n}_{s}_VR", "BLOB") for s in STATS]
    return cols


def test_descriptor_database_is_the_references_schema(tmp_path):
    """the sqlite file the host layer writes, read back with Python's sqlite3 + msgpack: user_version,
    the `assets` columns (names, order, declared types), a succeeded row, a failed row"""
    import sqlite3

    import msgpack
    from tests.golden.make_golden import column_values

    build()
    db = str(tmp_path / "afec-ll.db")
    out = subprocess.run([BIN, "sqlite", db], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    con = sqlite3.connect(db)
    assert con.execute("PRAGMA user_version").fetchone()[0] == 2
    info = con.execute("PRAGMA table_info(assets)").fetchall()
    assert [(r[1], r[2]) for r in info] == full_schema()
    assert len(info) == 3 + 6 + 4 + 22 * 14 + 2 * (14 + 6) + 2 + 7 * 14          # 461 columns
    assert [r[1] for r in info if r[5]] == ["filename"]                           # PRIMARY KEY
    rows = con.execute("SELECT filename, modtime, status FROM assets WHERE filename LIKE 'Kicks/%' ORDER BY filename").fetchall()
    assert rows == [("Kicks/broken.wav", 1700000002, "error: could not decode"), ("Kicks/one.wav", 1700000001, "succeeded")]
    # the rows written inside one transaction through the refilled column vector: each carries its own values
    batch = {r[0]: r for r in con.execute("SELECT filename, status, spectral_centroid_VR, cepstrum_bands_VVR, analyzation_offset_R, "
                                          "effectve_length_24dB_R FROM assets WHERE filename LIKE 'Batch/%'")}
    assert sorted(batch) == ["Batch/again.wav", "Batch/broken.wav", "Batch/long.wav", "Batch/noinfo.wav", "Batch/prepared.wav", "Batch/short.wav"]
    # InsertColumns (values built by another thread in the crawler) writes InsertSample's row
    cols = [r[1] for r in con.execute("PRAGMA table_info(assets)") if r[1] != "filename"]
    a_row = con.execute("SELECT %s FROM assets WHERE filename = 'Batch/again.wav'" % ", ".join(cols)).fetchone()
    p_row = con.execute("SELECT %s FROM assets WHERE filename = 'Batch/prepared.wav'" % ", ".join(cols)).fetchone()
    assert a_row == p_row
    assert batch["Batch/broken.wav"][1] == "error: could not decode" and batch["Batch/broken.wav"][2] is None
    assert msgpack.unpackb(batch["Batch/long.wav"][2]) == [0.5] * 40 and len(msgpack.unpackb(batch["Batch/long.wav"][3])) == 40
    assert msgpack.unpackb(batch["Batch/short.wav"][2]) == [7.0] and msgpack.unpackb(batch["Batch/short.wav"][3]) == []
    assert msgpack.unpackb(batch["Batch/again.wav"][2]) == column_values(5, 0, 1).tolist()
    assert msgpack.unpackb(batch["Batch/again.wav"][3]) == column_values(5, 14, 3).tolist()
    # without the load info the analysis offset is a column this library did not compute: the REAL placeholder
    assert batch["Batch/noinfo.wav"][4] == 0.0 and batch["Batch/again.wav"][4] != 0.0
    assert all(batch[k][5] == 1.25 for k in batch if k != "Batch/broken.wav")
    con.row_factory = sqlite3.Row
    ok = con.execute("SELECT * FROM assets WHERE filename = 'Kicks/one.wav'").fetchone()
    assert (ok["file_type_S"], ok["file_size_R"], ok["file_length_R"], ok["file_sample_rate_R"],
            ok["file_channel_count_R"], ok["file_bit_depth_R"]) == ("wav", 176444, 2.0, 44100, 1, 16)
    assert ok["effectve_length_24dB_R"] == 1.25 and ok["spectral_centroid_median_R"] == 2.5
    assert msgpack.unpackb(ok["spectral_centroid_VR"]) == column_values(5, 0, 1).tolist()
    assert msgpack.unpackb(ok["cepstrum_bands_VVR"]) == column_values(5, 14, 3).tolist()
    assert msgpack.unpackb(ok["cepstrum_bands_mean_VR"]) == [b / 4.0 for b in range(14)]
    # the rhythm tracker's columns (SampleDescriptors.cpp:180-195)
    assert ok["rhythm_final_tempo_R"] == 123.5 and ok["rhythm_final_tempo_confidence_R"] == 0.75
    assert msgpack.unpackb(ok["rhythm_complex_onsets_VR"]) == column_values(7, 0, 4).tolist()
    assert ok["rhythm_complex_onsets_max_R"] == 3.25 and ok["rhythm_percussive_onset_count_R"] == 4.0
    assert ok["rhythm_complex_onset_contrast_R"] == -0.125 and msgpack.unpackb(ok["rhythm_percussive_onsets_VR"]) == []
    # every column of a "succeeded" row is well formed, read back the way the reference reads it (every BLOB through
    # msgpack, SqliteSampleDescriptorPool.cpp:1004-1014)
    for name, kind in full_schema():
        if kind == "BLOB":
            assert isinstance(msgpack.unpackb(ok[name]), list), name
        elif kind == "REAL":
            assert isinstance(ok[name], float), name
    bad = con.execute("SELECT * FROM assets WHERE filename = 'Kicks/broken.wav'").fetchone()
    assert all(bad[k] is None for k in bad.keys() if k not in ("filename", "modtime", "status"))
    con.close()


def test_descriptor_database_version_rules(tmp_path):
    """TSqliteSampleDescriptorPool::InitializeDatabase (SqliteSampleDescriptorPool.cpp:1224-1300): a newer database is
    refused, an older one is thrown away and recreated, a current one is kept."""
    import sqlite3

    build()
    # newer: refused, untouched
    newer = str(tmp_path / "newer.db")
    con = sqlite3.connect(newer)
    con.execute("CREATE TABLE assets(filename TEXT PRIMARY KEY, x REAL)")
    con.execute("PRAGMA user_version = 3")
    con.commit(); con.close()
    out = subprocess.run([BIN, "sqlite", newer], capture_output=True, text=True)
    assert out.returncode != 0 and "Unknown database version" in (out.stdout + out.stderr)
    con = sqlite3.connect(newer)
    assert con.execute("PRAGMA user_version").fetchone()[0] == 3
    assert [r[1] for r in con.execute("PRAGMA table_info(assets)")] == ["filename", "x"]
    con.close()
    # older: dropped and recreated with the current schema
    older = str(tmp_path / "older.db")
    con = sqlite3.connect(older)
    con.execute("CREATE TABLE assets(filename TEXT PRIMARY KEY, old_column REAL)")
    con.execute("INSERT INTO assets VALUES ('stale.wav', 1.0)")
    con.execute("PRAGMA user_version = 1")
    con.commit(); con.close()
    out = subprocess.run([BIN, "sqlite", older], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    con = sqlite3.connect(older)
    assert con.execute("PRAGMA user_version").fetchone()[0] == 2
    assert [(r[1], r[2]) for r in con.execute("PRAGMA table_info(assets)")] == full_schema()
    assert con.execute("SELECT count(*) FROM assets WHERE filename = 'stale.wav'").fetchone()[0] == 0
    con.close()
    # current: rows of an earlier run survive a second run
    out = subprocess.run([BIN, "sqlite", older], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    con = sqlite3.connect(older)
    assert con.execute("SELECT count(*) FROM assets").fetchone()[0] == 8      # INSERT OR REPLACE: the same 8 files
    con.close()


def test_writer_rate_probe_runs(tmp_path):
    """`host_test writer_rate`: the database writer alone (no GPU), rows of a one-second stereo file; the numbers are
    quoted in HISTORY.md section 7.  Here only that it runs and that batching the commits does not change the rows."""
    import sqlite3
    build()
    dbs = []
    for per_txn in (0, 16):
        db = str(tmp_path / f"rate{per_txn}.db")
        out = subprocess.run([BIN, "writer_rate", db, "40", str(per_txn)], capture_output=True, text=True)
        assert out.returncode == 0 and "rows/s" in out.stdout, out.stdout + out.stderr
        con = sqlite3.connect(db)
        dbs.append(con.execute("SELECT * FROM assets ORDER BY filename").fetchall())
        con.close()
    assert len(dbs[0]) == 40 and dbs[0] == dbs[1]


@pytest.mark.gpu
def test_host_sample_analyser_matches_oracle():
    run("analyse")


def test_in_register_dft_blocks_against_a_direct_dft(tmp_path):
    """afx_fft32.h (dft32, dft16_rest, dft32_merge, the fused butterflies) and afx_fft.h's per-lane stages (radix4, dft16) on
    the host against a direct DFT in long double, and the 1024-point transform composed the way frames32_kernel composes
    it + the real-input untangle (tests/host/test_fft32_math.cpp; the file afx_fft32.h's header cites)."""
    exe = str(tmp_path / "test_fft32_math")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe,
                           os.path.join(ROOT, "tests", "host", "test_fft32_math.cpp"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all passed" in out.stdout and "FAILED" not in out.stdout
