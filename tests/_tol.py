"""Shared parity tolerances.

North-star bar: every descriptor within 1e-4 relative of the CPU reference.  An absolute floor is
needed where a descriptor is legitimately ~0 or is the difference of nearly equal numbers
(flatness-dB of a flat band = log of ~1; Pearson flux near the |denom|<=1e-12 cut; skew/kurt
which divide tiny magnitudes by the bin variance).  Floors are per descriptor and stated here.
"""
import numpy as np

# name -> (rtol, atol)
GPU_TOL = {
    "mfcc": (1e-4, 1e-6),
    "spectral_rms": (1e-4, 1e-12),
    "spectral_centroid": (1e-4, 1e-7),
    "spectral_spread": (1e-4, 1e-6),
    "spectral_skewness": (1e-4, 1e-9),
    "spectral_kurtosis": (1e-4, 1e-9),
    "spectral_rolloff": (0.0, 0.0),      # discrete (bins x 43): exact
    "spectral_flatness": (1e-4, 1e-6),
    "spectral_flux": (1e-4, 1e-7),
    "spectrum_bands": (1e-4, 1e-18),
    "sub_rms": (1e-4, 1e-12),
    "sub_flatness": (1e-4, 1e-6),
    "sub_flux": (1e-4, 1e-7),
    "sub_complexity": (0.0, 0.0),        # discrete count: exact
    "sub_contrast": (1e-4, 1e-9),
    "spectral_contrast": (1e-4, 1e-9),
    "amplitude_peak": (0.0, 0.0),        # exact (max of exactly representable inputs)
    "amplitude_rms": (1e-6, 1e-15),
}


def check(name, got, ref, rtol, atol, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.all(np.isfinite(got)), f"{what}{name}: non-finite output"
    err = np.abs(got - ref)
    lim = rtol * np.abs(ref) + atol
    bad = err > lim
    if np.any(bad):
        i = np.unravel_index(np.argmax(err - lim), err.shape)
        raise AssertionError(
            f"{what}{name}: {int(bad.sum())}/{bad.size} outside rtol={rtol} atol={atol}; worst at {i}: "
            f"got {got[i]!r} ref {ref[i]!r}")


def check_mag(got, ref, rel_to_max=1e-12, what=""):
    """Magnitudes: absolute error relative to the frame's largest bin."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    scale = np.maximum(ref.max(axis=-1, keepdims=True), 1e-300)
    err = np.abs(got - ref) / scale
    assert np.all(err <= rel_to_max), f"{what}mag: max err/max-bin = {err.max():.3e} > {rel_to_max}"


# ---- neighbours of the spectral set: (rtol, atol) against the oracle / reference goldens ----
# Flags and counts are exact.  f0 / confidence come from a 1024-lag correlation computed through FFTs on
# the GPU and by direct summation in the oracle (aubio: FFT); the values agree to ~1e-12 of the frame
# energy, the period after parabolic interpolation to ~1e-9.
NEIGH_TOL = {
    "amplitude_silence": (0.0, 0.0), "amplitude_envelope": (1e-9, 1e-15),
    "spectral_complexity": (0.0, 0.0), "auto_correlation": (1e-6, 1e-9),
    "f0": (1e-6, 1e-9), "f0_confidence": (1e-6, 1e-7), "failsafe_f0": (1e-6, 1e-9),
    "spectral_inharmonicity": (0.0, 0.0), "tristimulus1": (0.0, 0.0), "tristimulus2": (0.0, 0.0),
    "tristimulus3": (0.0, 0.0),
}
