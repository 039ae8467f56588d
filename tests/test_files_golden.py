"""Whole files at BASELINE's C3 / C4 shapes against goldens chained through the reference's own objects
(tests/golden/files.npz, made by tests/golden/make_golden_files.py: ref_driver load -> frames + neighbours): the
oracle chain on the CPU, and the GPU pipeline (LoadSample front end -> every per-frame descriptor) through the C-ABI."""
import os

import numpy as np
import pytest

from tests import _oracle, _tol
from tests._oracle import FIELDS, NEIGH_FIELDS, Oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "files.npz")
NAMES = ["c3_mono_2s_a", "c3_mono_2s_b", "c4_stereo_1s_a", "c4_stereo_1s_b"]
# the oracle restates the reference's arithmetic with another FFT: it agrees with the reference's objects to FFT rounding
ORACLE_RTOL = 1e-6


def spectral_fields():
    return [(f, a - 1024, b - 1024) for f, (a, b) in FIELDS.items() if f != "mag"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_chain_matches_the_reference_chain(name):
    z = np.load(GOLD)
    mono, info = _oracle.load_sample(z["raw_" + name], int(z["channels_" + name]))
    assert [info["data_offset"], info["silent_leading"], info["silent_trailing"], info["n_samples"]] == z["info_" + name].tolist()
    ora = Oracle()
    rec = ora.run(mono, cap=True)
    want = z["spectral_" + name]
    assert rec.shape[0] == want.shape[0]
    for field, a, b in spectral_fields():
        rtol, atol = _tol.GPU_TOL[field]
        _tol.check(field, rec[:, 1024 + a:1024 + b], want[:, a:b], min(rtol, ORACLE_RTOL) if rtol else 0.0, atol, what=f"{name} oracle ")
    nei = ora.run_neighbours(mono, cap=True)
    for field, col in NEIGH_FIELDS.items():
        rtol, atol = _tol.NEIGH_TOL[field]
        _tol.check(field, nei[:, col], z["neighbours_" + name][:, col], rtol, atol, what=f"{name} oracle ")


@pytest.mark.gpu
def test_gpu_pipeline_matches_the_reference_chain():
    import afec_amd as afx
    z = np.load(GOLD)
    files = [(z["raw_" + n], int(z["channels_" + n])) for n in NAMES]
    plan = afx.Plan()
    batch, infos = plan.batch_from_raw(files, afx.D_ALL_PER_FRAME)
    batch.run()
    res = batch.fetch()
    off = res["frame_offset"]
    for i, name in enumerate(NAMES):
        want_info = z["info_" + name].tolist()
        assert [infos[i]["data_offset"], infos[i]["silent_leading"], infos[i]["silent_trailing"], infos[i]["n_samples"]] == want_info
        assert infos[i]["peak_value"] == z["peakrms_" + name][0]
        want = z["spectral_" + name]
        assert off[i + 1] - off[i] == want.shape[0]
        for field, a, b in spectral_fields():
            rtol, atol = _tol.GPU_TOL[field]
            got = res[field][off[i]:off[i + 1]].reshape(want.shape[0], -1)
            _tol.check_gpu(field, got, want[:, a:b], rtol, atol, what=f"{name} ")
        for field, col in NEIGH_FIELDS.items():
            rtol, atol = _tol.NEIGH_TOL[field]
            _tol.check_gpu(field, res[field][off[i]:off[i + 1]], z["neighbours_" + name][:, col], rtol, atol, what=f"{name} ")
    batch.close()
    plan.close()
