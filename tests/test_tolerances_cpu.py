"""tests/_tol.py's regression ceiling: a loss of accuracy far inside the 1e-4 bar must fail check_gpu()."""
import numpy as np
import pytest

from tests import _tol


def test_a_hundredfold_loss_of_accuracy_passes_the_bar_and_fails_the_ceiling():
    rng = np.random.default_rng(3)
    ref = rng.uniform(-40.0, 40.0, (64, 14))                 # MFCC-like values
    good = ref * (1.0 + 9e-7 * rng.uniform(-1, 1, ref.shape))  # what the shipped kernel shows at worst (fuzz material)
    bad = ref * (1.0 + 6e-5 * rng.uniform(-1, 1, ref.shape))   # sixty times worse, still < 1e-4
    _tol.check("mfcc", bad, ref, *_tol.GPU_TOL["mfcc"])        # the north-star bar alone does not see it
    _tol.check_gpu("mfcc", good, ref)
    with pytest.raises(AssertionError, match="regression ceiling"):
        _tol.check_gpu("mfcc", bad, ref)
    assert _tol.over_ceiling("mfcc", good, ref) <= 1.0 < _tol.over_ceiling("mfcc", bad, ref)


def test_every_descriptor_has_a_ceiling_or_is_exact():
    for name in list(_tol.GPU_TOL) + list(_tol.NEIGH_TOL):
        rtol, atol = _tol.bar(name)
        if (rtol, atol) == (0.0, 0.0):
            assert name in _tol._EXACT, name
            assert _tol.over_ceiling(name, np.array([1.0]), np.array([1.0])) == 0.0
            assert _tol.over_ceiling(name, np.array([1.0 + 1e-15]), np.array([1.0])) == float("inf")
        else:
            assert _tol.OBSERVED_CEILING[name] <= rtol / 2.0, name       # every ceiling inside its bar


def test_record_mode_writes_the_worst_error_and_does_not_enforce(tmp_path, monkeypatch):
    import json
    path = tmp_path / "observed.json"
    monkeypatch.setenv("AFX_TOL_RECORD", str(path))
    ref = np.ones(8)
    _tol.check_gpu("spectral_rms", ref * (1 + 1e-6), ref)       # above the 2e-8 ceiling, inside the bar: recorded
    _tol.check_gpu("spectral_rms", ref * (1 + 1e-9), ref)
    got = json.load(open(path))
    assert abs(got["spectral_rms"] - 1e-6) < 1e-9
