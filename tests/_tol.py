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


# ---- regression ceiling on the OBSERVED error (round 5) ----
# The bars above are the north star's (1e-4 relative): three to eight orders of magnitude looser than what the f64
# kernels deliver.  A change that costs 100 x in accuracy (a shorter series for a logarithm, a reciprocal without its
# Newton step) would pass them all.  The ceilings below are 10 x the worst error the shipped kernels have shown, per
# descriptor, as the relative error |got - ref| / max(|ref|, atol / rtol) -- the metric of the parity report
# (profiles/r04/parity_report.md, profiles/r05/observed_errors.json: every check_gpu() call of one run of the whole
# GPU suite, tools/observed_errors.py); discrete descriptors stay exact.  check_gpu() asserts the bar AND the ceiling.
OBSERVED_CEILING = {
    # 10 x the worst error seen over one run of the whole GPU suite (profiles/r05/observed_errors.json) AND three fuzz soaks
    # of random material, masks and batch shapes (profiles/r05/observed_errors_fuzz.json + the soak of seed 51, which set
    # the sub-band and f0 entries): ragged random material reaches further than the suite's signals.
    # mfcc: logs of mel sums over leakage-floor bins (vector.c:350-391), the one descriptor whose error random material can
    # push far beyond the suite's (2.6e-7 there, 38 x under this ceiling): 9.8e-7 in round 5's soaks, 1.23e-5 ONCE in round
    # 6's (seed 93, round 3327, coefficient 7 of one frame; 1.45 M frames) -- 8 x inside the bar, above this ceiling, and
    # ten times it would not fit under the bar: the randomised runs report it as a warning (see f0 below).
    "mfcc": 1e-5,
    "sub_flatness": 2e-5,           # 1.75e-6 (fuzz; 3.6e-8 in the suite): geometric means of sub-bands of 2..6 bins (Statistics.cpp:417-455)
    "sub_contrast": 5e-7,           # 4.9e-8 (fuzz; 1.6e-9 in the suite): 10 x, no more
    # f0 / failsafe_f0: 4 x, not 10 x -- 1.25e-7 observed (fuzz: a parabolic interpolation over a nearly flat minimum; 2.6e-14
    # in the suite) and the bar itself is 1e-6: ten times the observed worst does not fit under it.  Randomised runs
    # (tests/fuzz_gpu.py, bench.py's spot checks) therefore report an excess over a ceiling as a WARNING with the seed
    # (ceiling_is_fatal=False below); the deterministic suite keeps the hard assert.
    "f0": 5e-7, "failsafe_f0": 5e-7,
    "spectral_skewness": 1e-6, "spectral_kurtosis": 1e-6,   # 8.5e-8 (fuzz, seed 72: a kurtosis of 0.006 = mean fourth power / sigma^4 - 3,
                                                            # two terms of size 3 cancelling; 7.6e-10 otherwise)
    "spectral_rms": 2e-8, "spectral_centroid": 2e-8, "spectral_spread": 2e-8,
    "spectral_flatness": 2e-8, "spectral_flux": 2e-8, "spectrum_bands": 2e-8,
    "sub_rms": 2e-8, "sub_flux": 2e-8, "spectral_contrast": 2e-8,          # all <= 1.6e-9 observed
    "amplitude_rms": 1e-12, "amplitude_envelope": 2e-12,        # time-domain sums of 1 024 samples: 1.4e-14 / 5e-15 observed
    "auto_correlation": 2e-8, "f0_confidence": 2e-8,           # 7.8e-11 / 8.7e-14 observed
}
_EXACT = {"spectral_rolloff", "sub_complexity", "amplitude_peak", "amplitude_silence", "spectral_complexity",
          "spectral_inharmonicity", "tristimulus1", "tristimulus2", "tristimulus3"}

_observed = {}      # name -> worst relative error seen by check_gpu() in this process


def bar(name):
    """(rtol, atol) of a descriptor: the north-star bar"""
    return GPU_TOL[name] if name in GPU_TOL else NEIGH_TOL[name]


def rel_err(name, got, ref):
    """|got - ref| / max(|ref|, floor), floor = atol / rtol of the descriptor's bar (0 / 0 = 0 for the exact ones)"""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    rtol, atol = bar(name)
    floor = atol / rtol if rtol > 0 else 0.0
    denom = np.maximum(np.abs(ref), floor) if floor > 0 else np.where(ref != 0, np.abs(ref), 1.0)
    return np.abs(got - ref) / denom


def over_ceiling(name, got, ref):
    """worst relative error in units of the descriptor's ceiling (<= 1 is inside; exact descriptors: 0 or inf)"""
    e = rel_err(name, got, ref)
    worst = float(e.max()) if e.size else 0.0
    if name in _EXACT:
        return 0.0 if worst == 0.0 else float("inf")
    return worst / OBSERVED_CEILING[name]


ceiling_warnings = []   # (what, name, worst, ceiling) of the excesses reported as warnings (ceiling_is_fatal=False)


def check_gpu(name, got, ref, rtol=None, atol=None, what="", ceiling_is_fatal=True):
    """The HIP path against the oracle / the reference's goldens: the 1e-4 bar (check) and the regression ceiling.
    AFX_TOL_RECORD=<file>: the worst relative error per descriptor of the process is written there (json) and the
    ceiling is not enforced -- how the ceilings were measured.  ceiling_is_fatal=False (randomised material: the fuzz
    soaks): an error inside the bar but above the ceiling is printed and kept in ceiling_warnings, not raised."""
    import json
    import os
    if rtol is None:
        rtol, atol = bar(name)
    check(name, got, ref, rtol, atol, what=what)
    e = rel_err(name, got, ref)
    worst = float(e.max()) if e.size else 0.0
    record = os.environ.get("AFX_TOL_RECORD")
    if record:
        if worst > _observed.get(name, -1.0):
            _observed[name] = worst
            try:
                old = json.load(open(record)) if os.path.exists(record) else {}
            except ValueError:
                old = {}
            old[name] = max(worst, old.get(name, 0.0))
            with open(record, "w") as f:
                json.dump(old, f, indent=1, sort_keys=True)
        return
    if name in _EXACT:
        return                                   # (0, 0) bars: check() has already demanded equality
    ceiling = OBSERVED_CEILING[name]
    if worst > ceiling:
        i = np.unravel_index(int(np.argmax(e)), e.shape)
        if not ceiling_is_fatal:
            import sys
            ceiling_warnings.append((what, name, worst, ceiling))
            print(f"WARNING {what}{name}: relative error {worst:.3e} at {i} is inside the {rtol:g} bar but above the regression "
                  f"ceiling {ceiling:g} (tests/_tol.py)", file=sys.stderr)
            return
        raise AssertionError(
            f"{what}{name}: relative error {worst:.3e} at {i} is inside the {rtol:g} bar but above the regression ceiling "
            f"{ceiling:g} (10 x the worst error the shipped kernels have shown, tests/_tol.py): got "
            f"{np.asarray(got, dtype=np.float64)[i]!r} ref {np.asarray(ref, dtype=np.float64)[i]!r}")
