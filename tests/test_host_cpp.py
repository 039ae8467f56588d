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
    assert len(order) == 4 + 22 * 14 + 7 * 14
    assert cols["effectve_length_48dB_R"] == 1.5 and cols["effectve_length_12dB_R"] == 0.5
    # SamplesToMs in float, then / 1000 (SampleAnalyser.cpp:748-749)
    assert cols["analyzation_offset_R"] == float(np.float32(-2205) / (np.float32(44100) / np.float32(1000))) / 1000.0
    assert cols["spectral_centroid_VR"] == column_values(5, 0, 1).tolist()
    assert cols["spectral_centroid_median_R"] == 2.5 and cols["spectral_centroid_dvariance_R"] == -0.125
    assert cols["f0_VR"] == column_values(5, 0, 2).tolist()
    assert cols["cepstrum_bands_VVR"] == column_values(5, 14, 3).tolist()
    assert cols["cepstrum_bands_mean_VR"] == [b / 4.0 for b in range(14)]
    assert cols["amplitude_peak_VR"] == [] and cols["frequency_bands_VVR"] == []


@pytest.mark.gpu
def test_host_sample_analyser_matches_oracle():
    run("analyse")
