"""ctypes wrapper around oracle/libafx_oracle.so (the CPU parity checker).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (afec_amd) never imports this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
RECORD = 1147
# record layout (oracle/afx_oracle.h)
FIELDS = {
    "mag": (0, 1024), "mfcc": (1024, 1038), "spectral_rms": (1038, 1039),
    "spectral_centroid": (1039, 1040), "spectral_spread": (1040, 1041),
    "spectral_skewness": (1041, 1042), "spectral_kurtosis": (1042, 1043),
    "spectral_rolloff": (1043, 1044), "spectral_flatness": (1044, 1045),
    "spectral_flux": (1045, 1046), "spectrum_bands": (1046, 1074),
    "sub_rms": (1074, 1088), "sub_flatness": (1088, 1102), "sub_flux": (1102, 1116),
    "sub_complexity": (1116, 1130), "sub_contrast": (1130, 1144),
    "spectral_contrast": (1144, 1145), "amplitude_peak": (1145, 1146),
    "amplitude_rms": (1146, 1147),
}

NEIGH_RECORD = 1035
# neighbours record layout (oracle/afx_oracle.h AFXN_*)
NEIGH_FIELDS = {
    "amplitude_silence": 0, "amplitude_envelope": 1, "f0": 2, "f0_confidence": 3, "failsafe_f0": 4,
    "auto_correlation": 5, "spectral_complexity": 6, "spectral_inharmonicity": 7,
    "tristimulus1": 8, "tristimulus2": 9, "tristimulus3": 10,
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libafx_oracle.so")
        src = os.path.join(ORACLE_DIR, "afx_oracle.c")
        srcs = [src, os.path.join(ORACLE_DIR, "afx_oracle_rhythm.c"), os.path.join(ORACLE_DIR, "afx_oracle.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libafx_oracle.so"],
                                  stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        L.afx_oracle_create.restype = ctypes.c_void_p
        L.afx_oracle_create.argtypes = [ctypes.c_int] * 3
        L.afx_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.afx_oracle_window.restype = ctypes.POINTER(ctypes.c_double)
        L.afx_oracle_window.argtypes = [ctypes.c_void_p]
        L.afx_oracle_mel.restype = ctypes.POINTER(ctypes.c_double)
        L.afx_oracle_mel.argtypes = [ctypes.c_void_p]
        L.afx_oracle_first_bin.argtypes = [ctypes.c_void_p]
        L.afx_oracle_bin_count.argtypes = [ctypes.c_void_p]
        L.afx_oracle_num_frames.restype = ctypes.c_int64
        L.afx_oracle_num_frames.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        L.afx_oracle_run.restype = ctypes.c_int64
        L.afx_oracle_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                     ctypes.c_void_p]
        L.afx_oracle_run_neighbours.restype = ctypes.c_int64
        L.afx_oracle_run_neighbours.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_void_p]
        L.afx_oracle_peaks.restype = ctypes.c_int
        L.afx_oracle_peaks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                       ctypes.c_void_p]
        L.afx_oracle_effective_length.restype = None
        L.afx_oracle_effective_length.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.afx_oracle_run_mfcc.restype = ctypes.c_int64
        L.afx_oracle_run_mfcc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_void_p]
        dp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        for name, args in [("sum", [dp, ci]), ("mean", [dp, ci]), ("variance", [dp, ci, cd]),
                           ("geometric_mean", [dp, ci]), ("centroid", [dp, ci]),
                           ("spread", [dp, ci, cd]), ("skewness", [dp, ci, cd, cd]),
                           ("kurtosis", [dp, ci, cd, cd]), ("flatness", [dp, ci]),
                           ("flatness_db", [dp, ci]), ("correlation", [dp, dp, ci]),
                           ("median", [dp, ci]), ("min", [dp, ci]), ("max", [dp, ci]),
                           ("lin_to_db", [cd])]:
            fn = getattr(L, "afx_oracle_" + name)
            fn.restype = ctypes.c_double
            fn.argtypes = args
        L.afx_oracle_calc_statistics.argtypes = [dp, ci, dp]
        _lib = L
    return _lib


class _LoadInfo(ctypes.Structure):
    _fields_ = [("peak_value", ctypes.c_float), ("rms_value", ctypes.c_float), ("data_offset", ctypes.c_int32),
                ("silent_leading", ctypes.c_int32), ("silent_trailing", ctypes.c_int32), ("n_samples", ctypes.c_int64)]


def load_sample(data, channels, fft=2048, file_rate=44100, rate=44100):
    """oracle LoadSample: data int16 / float32 interleaved, or uint8 packed int24 -> (mono doubles, info dict).
    file_rate != rate: the mono mix is resampled first (SA:563-607)."""
    L = lib()
    L.afx_oracle_load_sample_at.restype = ctypes.c_void_p
    L.afx_oracle_load_sample_at.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.POINTER(_LoadInfo)]
    L.afx_oracle_free.argtypes = [ctypes.c_void_p]
    data = np.ascontiguousarray(np.asarray(data).reshape(-1))
    if data.dtype == np.int16:
        fmt, frames = 0, data.size // channels
    elif data.dtype == np.uint8:
        fmt, frames = 1, data.size // (3 * channels)
    elif data.dtype == np.int32:
        fmt, frames = 3, data.size // channels
    elif data.dtype == np.float64:
        fmt, frames = 4, data.size // channels
    else:
        data = data.astype(np.float32)
        fmt, frames = 2, data.size // channels
    info = _LoadInfo()
    p = L.afx_oracle_load_sample_at(data.ctypes.data, fmt, channels, frames, file_rate, rate, fft, ctypes.byref(info))
    out = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), (info.n_samples,)).copy()
    L.afx_oracle_free(p)
    return out, {k: getattr(info, k) for k, _ in _LoadInfo._fields_}


def resample(x, file_rate, rate=44100):
    """oracle libresample as LoadSample drives it (SA:563-607): float32 mono in -> (float32 out, samples the converter wrote)."""
    L = lib()
    L.afx_oracle_resample.restype = ctypes.c_void_p
    L.afx_oracle_resample.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int64)]
    L.afx_oracle_free.argtypes = [ctypes.c_void_p]
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_out, n_written = ctypes.c_int64(0), ctypes.c_int64(0)
    p = L.afx_oracle_resample(x.ctypes.data, x.size, file_rate, rate, ctypes.byref(n_out), ctypes.byref(n_written))
    out = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), (n_out.value,)).copy()
    L.afx_oracle_free(p)
    return out, n_written.value


def resample_filter():
    L = lib()
    imp = np.zeros(4096 * 17, dtype=np.float32)
    L.afx_oracle_resample_filter.argtypes = [ctypes.c_void_p]
    L.afx_oracle_resample_filter(imp.ctypes.data)
    return imp


class Oracle:
    def __init__(self, sample_rate=44100, fft=2048, hop=1024):
        self.L = lib()
        self.fft, self.hop = fft, hop
        self.h = ctypes.c_void_p(self.L.afx_oracle_create(sample_rate, fft, hop))
        assert self.h.value

    def __del__(self):
        try:
            self.L.afx_oracle_destroy(self.h)
        except Exception:
            pass

    def window(self):
        return np.ctypeslib.as_array(self.L.afx_oracle_window(self.h), (self.fft,)).copy()

    def mel(self):
        return np.ctypeslib.as_array(self.L.afx_oracle_mel(self.h), (14 * (self.fft // 2),)).reshape(14, -1).copy()

    def first_bin(self):
        return self.L.afx_oracle_first_bin(self.h)

    def bin_count(self):
        return self.L.afx_oracle_bin_count(self.h)

    def num_frames(self, n, cap=False):
        return int(self.L.afx_oracle_num_frames(self.h, int(n), int(cap)))

    def run(self, x, cap=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        nf = self.num_frames(x.size, cap)
        out = np.zeros((nf, RECORD), dtype=np.float64)
        if nf:
            self.L.afx_oracle_run(self.h, x.ctypes.data, x.size, int(cap), out.ctypes.data)
        return out

    def run_neighbours(self, x, cap=False):
        """[frames][NEIGH_RECORD]: the 11 scalars of NEIGH_FIELDS, then the whitened spectrum [1024]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        nf = self.num_frames(x.size, cap)
        out = np.zeros((nf, NEIGH_RECORD), dtype=np.float64)
        if nf:
            self.L.afx_oracle_run_neighbours(self.h, x.ctypes.data, x.size, int(cap), out.ctypes.data)
        return out

    def effective_length(self, x):
        """seconds above -48 / -24 / -12 dB (CalcEffectiveLength)"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(3)
        self.L.afx_oracle_effective_length(self.h, x.ctypes.data, x.size, out.ctypes.data)
        return out

    def rhythm_frames(self, n, cap=False):
        self.L.afx_oracle_rhythm_frames.restype = ctypes.c_int64
        self.L.afx_oracle_rhythm_frames.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        return int(self.L.afx_oracle_rhythm_frames(self.h, int(n), int(cap)))

    def run_rhythm(self, x, original_rate=44100, original_samples=None, data_offset=0, cap=False):
        """The 512/128 rhythm tracker loop: dict with onsets [2][T], sharpened [2][T], odf [2][T] and the 14 scalars
        (RHYTHM_SCALARS order)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        t = self.rhythm_frames(x.size, cap)
        onsets, sharp, odf = (np.zeros((2, t)) for _ in range(3))
        out = np.zeros(14)
        self.L.afx_oracle_run_rhythm.restype = ctypes.c_int64
        self.L.afx_oracle_run_rhythm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 4
        self.L.afx_oracle_run_rhythm(self.h, x.ctypes.data, x.size, int(cap), int(original_rate),
                                     int(x.size if original_samples is None else original_samples), int(data_offset),
                                     onsets.ctypes.data, sharp.ctypes.data, odf.ctypes.data, out.ctypes.data)
        return {"onsets": onsets, "sharpened": sharp, "odf": odf, "scalars": out}

    def run_mfcc(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        nf = self.num_frames(x.size, False)
        out = np.zeros((nf, 14), dtype=np.float64)
        if nf:
            self.L.afx_oracle_run_mfcc(self.h, x.ctypes.data, x.size, out.ctypes.data)
        return out


RHYTHM_SCALARS = [f"rhythm_{k}_{n}" for k in ("complex", "percussive")
                  for n in ("onset_count", "tempo", "tempo_confidence", "onset_frequency_mean", "onset_strength",
                            "onset_contrast")] + ["rhythm_final_tempo", "rhythm_final_tempo_confidence"]


def onset_polar(x512):
    """TOnsetFftProcessor::LoadFrame on one frame -> float32 [2 + 255 + 255]: dc, nyquist, magnitudes, phases."""
    x = np.ascontiguousarray(x512, dtype=np.float64)
    assert x.size == 512
    out = np.zeros(512, dtype=np.float32)
    L = lib()
    L.afx_oracle_onset_polar.restype = None
    L.afx_oracle_onset_polar.argtypes = [ctypes.c_void_p] * 5
    b = out.ctypes.data
    L.afx_oracle_onset_polar(x.ctypes.data, b, b + 4, b + 8, b + 8 + 4 * 255)
    return out


def beattrack(df, hop=128, rate=44100):
    """one aubio beat-tracking pass on an onset series -> (bpm, confidence)"""
    df = np.ascontiguousarray(df, dtype=np.float64)
    bpm, conf = ctypes.c_double(), ctypes.c_double()
    L = lib()
    L.afx_oracle_beattrack.restype = None
    L.afx_oracle_beattrack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p]
    L.afx_oracle_beattrack(df.ctypes.data, df.size, hop, rate, ctypes.byref(bpm), ctypes.byref(conf))
    return bpm.value, conf.value


def canny(x):
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    L = lib()
    L.afx_oracle_canny.restype = None
    L.afx_oracle_canny.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.afx_oracle_canny(x.ctypes.data, x.size)
    return x


def _vec(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def stat(name, *arrays_and_scalars):
    """Call afx_oracle_<name>(array..., n, scalars...)."""
    L = lib()
    arrs = [_vec(a) for a in arrays_and_scalars if isinstance(a, (list, tuple, np.ndarray))]
    scal = [a for a in arrays_and_scalars if not isinstance(a, (list, tuple, np.ndarray))]
    fn = getattr(L, "afx_oracle_" + name)
    if not arrs:
        return fn(*scal)
    n = arrs[0].size
    ptrs = [a.ctypes.data if a.size else None for a in arrs]
    return fn(*ptrs, n, *scal)


def calc_statistics(x, init=None):
    x = _vec(x)
    out = np.zeros(13) if init is None else _vec(init).copy()
    lib().afx_oracle_calc_statistics(x.ctypes.data if x.size else None, x.size, out.ctypes.data)
    return out


def peaks(x, threshold):
    """TStatistics::Peaks restatement -> list of (bin, value)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    bins = np.zeros(max(x.size, 1), dtype=np.int32)
    vals = np.zeros(max(x.size, 1), dtype=np.float64)
    c = lib().afx_oracle_peaks(x.ctypes.data, x.size, float(threshold), bins.ctypes.data, vals.ctypes.data)
    return [(int(bins[i]), float(vals[i])) for i in range(c)]
