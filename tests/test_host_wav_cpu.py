"""The host WAV reader (afec_amd/host/WaveFile.cpp: the read side of the reference's TWaveFile, WaveFile.cpp:365-410,
RiffFile.cpp:176-228) and the crawl's shard assignment.  CPU only.  The payloads of the golden WAVs are pushed
through the oracle's LoadSample and compared with what the reference's own converters produced
(tests/golden/load_wav.npz, from oracle/_ref/ref_driver)."""
import os
import struct

import numpy as np
import pytest

from tests import _host, _oracle
from tests._wav import wav_bytes

GOLD = os.path.join(os.path.dirname(__file__), "golden", "load_wav.npz")
NAMES = ["u8_mono", "i16_stereo", "i24_mono", "i32_stereo", "f32_mono", "f64_stereo"]
# TWaveFile::TSampleType in afec_amd/host/WaveFile.h
SAMPLE_TYPE = {"u8_mono": 0, "i16_stereo": 1, "i24_mono": 2, "i32_stereo": 3, "f32_mono": 4, "f64_stereo": 5}
RAW_FORMAT = {"u8_mono": 0, "i16_stereo": 0, "i24_mono": 1, "i32_stereo": 3, "f32_mono": 2, "f64_stereo": 4}


@pytest.mark.parametrize("name", NAMES)
def test_reader_and_oracle_against_the_reference_converters(name):
    z = np.load(GOLD)
    image = z["wav_" + name].tobytes()
    props, payload = _host.wave_probe(image)
    channels, rate, bits, frames = [int(v) for v in z["props_" + name]]
    assert (props["channels"], props["rate"], props["bits"], props["frames"]) == (channels, rate, bits, frames)
    assert props["sample_type"] == SAMPLE_TYPE[name] and props["raw_format"] == RAW_FORMAT[name]
    dtype = {0: np.int16, 1: np.uint8, 2: np.float32, 3: np.int32, 4: np.float64}[props["raw_format"]]
    data = np.frombuffer(payload, dtype=dtype)
    mono, info = _oracle.load_sample(data, channels)
    want_info = z["info_" + name]
    assert [info["data_offset"], info["silent_leading"], info["silent_trailing"], info["n_samples"]] == want_info.tolist()
    np.testing.assert_array_equal(mono, z["data_" + name])       # bit-exact: conversions, mix, normalisation, trim
    pr = z["peakrms_" + name]
    assert info["peak_value"] == pr[0] and abs(info["rms_value"] - pr[1]) <= 1e-6 * pr[1]


def test_files_the_reference_rejects():
    """UnitTests.cpp:338-350 expects exactly one failed sample in its test set: "_Not A Wavefile.wav"."""
    ok = wav_bytes(np.zeros(100, np.int16), 1, 16)
    for image, text in [
        (b"this is not a wave file, just some text that is long enough to be walked as chunks........", "Not a valid WAV file."),
        (b"", "Not a valid WAV file."),
        (ok[:12] + ok[36:], "Not a valid WAV file."),                                  # no fmt chunk
        (ok.replace(b"data", b"dat_"), "Not a valid WAV file."),                       # no data chunk
        (ok[:20] + struct.pack("<H", 2) + ok[22:], "Unsupported file format."),        # ADPCM format tag
        (ok[:28] + struct.pack("<I", 12345) + ok[32:], "Unsupported file format."),    # AvgBytesPerSec does not match
        # an empty data chunk at the very end is not even visited by the chunk walk (RiffFile.cpp:206-207: strict <)
        (wav_bytes(np.zeros(0, np.int16), 1, 16), "Not a valid WAV file."),
        # an empty chunk in front of `data` ends the chunk walk there (RiffFile.cpp:206: the next header must lie behind
        # the position after this one), so `data` is never found
        (ok[:36] + b"JUNK" + struct.pack("<I", 0) + ok[36:], "Not a valid WAV file."),
        # a sampling rate of 0 (the byte-rate check alone would pass: 0 == 0)
        (ok[:24] + struct.pack("<II", 0, 0) + ok[32:], "Unsupported file format."),
        # a data chunk shorter than one sample frame
        (wav_bytes(np.zeros(3, np.uint8), 2, 16)[:36] + b"data" + struct.pack("<I", 3) + b"\x00\x00\x00\x00", "Unsupported file format or corrupt file."),
    ]:
        with pytest.raises(RuntimeError) as ei:
            _host.wave_probe(image)
        assert str(ei.value) == text, (str(ei.value), text)


def test_odd_chunks_extensible_headers_and_truncated_data():
    x = (np.arange(2000) % 251 - 125).astype(np.int16)
    plain, _ = _host.wave_probe(wav_bytes(x, 2, 16))
    for kw in (dict(extra_chunks=True), dict(extensible=True), dict(extra_chunks=True, extensible=True)):
        props, payload = _host.wave_probe(wav_bytes(x, 2, 16, **kw))
        assert props == plain and payload == x.tobytes()
    cut = wav_bytes(x, 2, 16)[:-1000]                 # a data chunk that claims more than the file holds
    props, payload = _host.wave_probe(cut)
    assert props["frames"] == (len(cut) - 44) // 4 and payload == x.tobytes()[:props["frames"] * 4]

def test_files_on_disk_read_like_their_images(tmp_path):
    """TWaveFile::OpenForRead(FileName) parses the file's head and reads the data chunk by pread (what the crawler does
    for files on disk: straight into its staging buffer); it must see exactly what the image reader sees -- all sample
    types, a data chunk behind the 4 KiB head (a large chunk in front of it), files shorter than the head, and the
    same error texts."""
    z = np.load(GOLD)
    images = {name: z["wav_" + name].tobytes() for name in NAMES}
    x = (np.arange(30000) % 251 - 125).astype(np.int16)
    plain = wav_bytes(x, 2, 16)
    images["big_chunk_first"] = plain[:36] + b"LIST" + struct.pack("<I", 10000) + bytes(10000) + plain[36:]
    images["u8_behind_head"] = wav_bytes(((np.arange(9001) * 7) % 256).astype(np.uint8), 1, 8)
    images["tiny"] = wav_bytes(x[:10], 1, 16)
    images["truncated"] = plain[:-1000]
    fixtures = os.path.join(os.path.dirname(__file__), "golden", "wav")
    for f in sorted(os.listdir(fixtures)):
        if f.endswith(".wav") and not f.startswith("_"):
            images["fixture_" + f] = open(os.path.join(fixtures, f), "rb").read()
    for name, image in images.items():
        path = tmp_path / (name + ".wav")
        path.write_bytes(image)
        assert _host.wave_probe_file(path) == _host.wave_probe(image), name
    for k, image in enumerate([b"", b"RIFF", b"this is not a wave file, just some text that is long enough to be walked as chunks........",
                               plain.replace(b"data", b"dat_"), plain[:20] + struct.pack("<H", 2) + plain[22:]]):
        path = tmp_path / f"bad{k}.wav"
        path.write_bytes(image)
        with pytest.raises(RuntimeError) as from_file:
            _host.wave_probe_file(path)
        with pytest.raises(RuntimeError) as from_image:
            _host.wave_probe(image)
        assert str(from_file.value) == str(from_image.value)
    with pytest.raises(RuntimeError) as ei:
        _host.wave_probe_file(tmp_path / "missing.wav")
    assert str(ei.value) == f"Failed to open the file '{tmp_path / 'missing.wav'}'."
    with pytest.raises(RuntimeError):
        _host.wave_probe_file(tmp_path)                       # a directory


def test_shard_assignment():
    """File i of a crawl goes to device i mod G (one self-contained task per file, Crawler.cpp:706-728)."""
    L = _host.lib()
    for g in (1, 2, 3, 8):
        counts = [0] * g
        for i in range(1000):
            d = L.afec_shard_of_file(i, g)
            assert d == i % g
            counts[d] += 1
        assert max(counts) - min(counts) <= 1
