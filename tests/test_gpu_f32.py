"""AFX_PRECISION_F32 (float butterflies, double descriptor accumulation): meets the 1e-4 bar on
broadband material, where every analysed bin is far above the f32 FFT's error floor
(~1e-7 x the frame's largest bin).  Pure tones are the documented exception:
their leakage-floor bins sit below that floor, and the f64 mode exists for them."""
import numpy as np
import pytest

import afec_amd as afx
from tests import _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu

# discrete outputs may flip by one step when the deciding comparison is closer than f32 noise
DISCRETE = {"spectral_rolloff": 43.0, "sub_complexity": 1.0}


def broadband_signals():
    rng = np.random.default_rng(99)
    n = 2048 + 1024 * 63
    t = np.arange(n)
    white = rng.uniform(-1, 1, n)
    pink = np.cumsum(rng.standard_normal(n)) * 0.01
    pink -= np.mean(pink)
    pink /= np.max(np.abs(pink))
    mix = 0.3 * np.sin(2 * np.pi * 220 * t / 44100) + 0.2 * np.sin(2 * np.pi * 3100 * t / 44100) + 0.2 * white
    burst = white * np.exp(-t / 20000.0)
    return {"white": white, "brown": pink, "mix": mix, "burst": burst}


@pytest.mark.parametrize("name", sorted(broadband_signals()))
def test_f32_mode_on_broadband(name):
    x = broadband_signals()[name].astype(np.float32)
    plan = afx.Plan(precision=afx.PRECISION_F32, max_analysis_ms=0)
    res = plan.extract([x], afx.D_ALL_LOW_LEVEL)
    plan.close()
    ref = Oracle().run(x.astype(np.float64))
    for field, (a, b) in FIELDS.items():
        if field not in res:
            continue
        got = res[field].reshape(ref.shape[0], -1)
        want = ref[:, a:b]
        if field in DISCRETE:
            diff = np.abs(got - want)
            assert np.all(diff <= DISCRETE[field]), (field, diff.max())
            assert np.mean(diff > 0) < 0.02, (field, np.mean(diff > 0))
            continue
        rtol, atol = _tol.GPU_TOL[field]
        # skew/kurt divide (x - centroid) by the bin variance: amplification of the f32 magnitude
        # error by |centroid/spread| ~ 1e2 is inherent to the formula
        if field in ("spectral_skewness", "spectral_kurtosis", "sub_flatness", "spectral_flatness", "sub_contrast",
                     "spectral_contrast", "sub_flux", "spectral_flux"):
            atol = max(atol, 2e-5)
        # an MFCC is a signed sum of 14 log-energies of magnitude ~10: 1e-7 relative error on each
        # energy is ~1e-6 absolute on coefficients that happen to cancel to ~0
        if field == "mfcc":
            atol = 2e-5
        _tol.check(field, got, want, rtol, atol, what=f"f32 {name} ")


def test_f32_mode_c2_mfcc_on_benchmark_input():
    rng = np.random.default_rng(1234)
    x = rng.uniform(-1, 1, 2048 + 1024 * 999).astype(np.float32)
    plan = afx.Plan(precision=afx.PRECISION_F32, max_analysis_ms=0)
    res = plan.extract([x], afx.D_C2)
    plan.close()
    want = Oracle().run_mfcc(x.astype(np.float64))
    _tol.check("mfcc", res["mfcc"], want, 1e-4, 2e-5, what="f32 C2 ")
