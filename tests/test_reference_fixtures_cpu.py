"""The oracle against the reference's own objects on ALL of the reference's fixture WAVs (75 files under
Source/Crawler/XUnitTests/Resources/Kicks-vs-Snare-{Train,Test}: the folders UnitTests.cpp:152-423 crawls), computed on
the spot: oracle/_ref/ref_driver (built from the reference's sources by `make -C oracle ref`) runs LoadSample's
converters, the per-frame spectral loop and its neighbours on every file, and the oracle must agree -- LoadSample bit
for bit, the descriptors to 1e-6 (another FFT algorithm).  tests/golden/ holds thirteen of these files with their
expected values for the GPU box; this test needs the reference checkout and skips without it.  CPU only."""
import glob
import os

import numpy as np
import pytest

from tests import _oracle, _tol
from tests._oracle import FIELDS, NEIGH_FIELDS, Oracle
from tests._wav import parse_wav

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
RESOURCES = "/root/reference/Source/Crawler/XUnitTests/Resources"
FILES = sorted(glob.glob(os.path.join(RESOURCES, "**", "*.[wW][aA][vV]"), recursive=True))

pytestmark = pytest.mark.skipif(not FILES or not os.path.exists(REF), reason="needs /root/reference and oracle/_ref/ref_driver (make -C oracle ref)")


def test_every_fixture_of_the_reference():
    from tests.golden.make_golden import run_ref
    from tests.golden.make_golden_wav import run_ref_load
    ora = Oracle()
    checked = frames = rejected = 0
    for path in FILES:
        name = os.path.relpath(path, RESOURCES)
        try:
            channels, rate, bits, n, payload = parse_wav(open(path, "rb").read())
        except ValueError:
            rejected += 1                                   # "_Not A Wavefile.wav": UnitTests.cpp:338-350 expects exactly this one to fail
            continue
        assert rate == 44100 and bits in (16, 24), name
        data = np.frombuffer(payload, dtype=np.int16 if bits == 16 else np.uint8)
        peakrms, info_ref, mono_ref = run_ref_load(payload, 0 if bits == 16 else 1, channels, n)
        mono, info = _oracle.load_sample(data, channels)
        assert [info["data_offset"], info["silent_leading"], info["silent_trailing"], info["n_samples"]] == info_ref.tolist(), name
        np.testing.assert_array_equal(mono.view(np.uint64), mono_ref.view(np.uint64), err_msg=name)
        assert np.float32(info["peak_value"]) == peakrms[0] and abs(info["rms_value"] - peakrms[1]) <= 1e-6 * peakrms[1], name
        want = run_ref([mono], cap=1)
        rec = ora.run(mono, cap=True)
        assert rec.shape[0] == want.shape[0], name
        for field, (a, b) in FIELDS.items():
            if field == "mag":
                continue
            rtol, atol = _tol.GPU_TOL[field]
            _tol.check(field, rec[:, a:b], want[:, a:b], min(rtol, 1e-6) if rtol else 0.0, atol, what=f"{name} oracle ")
        want_n = run_ref([mono], cap=1, mode="neighbours", record=_oracle.NEIGH_RECORD)
        nei = ora.run_neighbours(mono, cap=True)
        for field, col in NEIGH_FIELDS.items():
            rtol, atol = _tol.NEIGH_TOL[field]
            _tol.check(field, nei[:, col], want_n[:, col], rtol, atol, what=f"{name} oracle ")
        checked += 1
        frames += rec.shape[0]
    assert rejected == 1 and checked == len(FILES) - 1 and checked >= 70 and frames > 500
