"""ctypes binding of include/afx.h (libafx_hip.so).

The library is the only compute path: if it is missing or fails to load this module raises --
there is no Python or CPU fallback.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFX_LIBRARY selects an alternative build of the same library (kernel tuning experiments)
_LIB_PATH = os.environ.get("AFX_LIBRARY") or os.path.join(_HERE, "lib", "libafx_hip.so")

D_MFCC = 1 << 0
D_SPECTRAL_RMS = 1 << 1
D_SPECTRAL_CENTROID = 1 << 2
D_SPECTRAL_SPREAD = 1 << 3
D_SPECTRAL_SKEWNESS = 1 << 4
D_SPECTRAL_KURTOSIS = 1 << 5
D_SPECTRAL_ROLLOFF = 1 << 6
D_SPECTRAL_FLATNESS = 1 << 7
D_SPECTRAL_FLUX = 1 << 8
D_SPECTRUM_BANDS = 1 << 9
D_BAND_FEATURES = 1 << 10
D_AMPLITUDE_PEAK = 1 << 11
D_AMPLITUDE_RMS = 1 << 12
D_MAGNITUDE = 1 << 13
D_STATISTICS = 1 << 14
# neighbours of the spectral set
D_AMPLITUDE_SILENCE = 1 << 15
D_AMPLITUDE_ENVELOPE = 1 << 16
D_SPECTRAL_COMPLEXITY = 1 << 17
D_AUTO_CORRELATION = 1 << 18
D_F0 = 1 << 19
D_SPECTRAL_INHARMONICITY = 1 << 20
D_TRISTIMULUS = 1 << 21
D_EFFECTIVE_LENGTH = 1 << 22   # per buffer: [n_bufs][3]
D_RHYTHM = 1 << 23             # per buffer: the 512/128 rhythm tracker (fetch_rhythm)
RHYTHM_SCALARS = [f"rhythm_{k}_{n}" for k in ("complex", "percussive")
                  for n in ("onset_count", "tempo", "tempo_confidence", "onset_frequency_mean", "onset_strength",
                            "onset_contrast")] + ["rhythm_final_tempo", "rhythm_final_tempo_confidence"]
NUM_STATISTICS = 13
STAT_NAMES = ["min", "max", "median", "mean", "gmean", "variance", "centroid", "spread", "skewness",
              "kurtosis", "flatness", "dmean", "dvariance"]
D_C2 = D_MFCC
D_SPECTRAL_STATS = 0x1FE
D_ALL_LOW_LEVEL = 0x1FFF
D_NEIGHBOURS = 0x3F8000
D_ALL_PER_FRAME = D_ALL_LOW_LEVEL | D_NEIGHBOURS
PRECISION_F64, PRECISION_F32 = 0, 1
PCM_F32, PCM_F64 = 0, 1
FRAME_KERNEL_AUTO, FRAME_KERNEL_WAVE64, FRAME_KERNEL_HALFWAVE = 0, 1, 2   # afx_plan_desc.frame_kernel
PLAN_NO_SIDE_STREAM = 1                                                   # afx_plan_desc.flags

# every symbol include/afx.h declares (tests check the library exports exactly these)
EXPORTS = [
    "afx_status_str", "afx_last_error", "afx_build_info", "afx_plan_create", "afx_plan_destroy",
    "afx_plan_get_window", "afx_plan_get_mel_table", "afx_plan_get_bin_range", "afx_num_frames",
    "afx_extract_batch", "afx_batch_create", "afx_batch_total_frames", "afx_batch_run",
    "afx_batch_sync", "afx_batch_run_timed", "afx_batch_fetch", "afx_batch_fetch_statistics", "afx_batch_destroy",
    "afx_algorithmic_bytes_per_frame", "afx_batch_create_from_raw", "afx_batch_fetch_samples",
    "afx_host_alloc", "afx_host_free", "afx_batch_record_layout", "afx_batch_fetch_records",
    "afx_batch_set_file_info", "afx_batch_rhythm_frames", "afx_batch_fetch_rhythm", "afx_batch_fetch_onset_functions",
    "afx_plan_set_blocking_wait", "afx_batch_get_info", "afx_plan_probe_device", "afx_device_count",
]
RAW_I16, RAW_I24, RAW_F32, RAW_I32, RAW_F64 = 0, 1, 2, 3, 4

# afx_out fields: name -> width per frame, in declaration order
OUT_FIELDS = [
    ("mfcc", 14), ("spectral_rms", 1), ("spectral_centroid", 1), ("spectral_spread", 1),
    ("spectral_skewness", 1), ("spectral_kurtosis", 1), ("spectral_rolloff", 1),
    ("spectral_flatness", 1), ("spectral_flux", 1), ("spectrum_bands", 28), ("sub_rms", 14),
    ("sub_flatness", 14), ("sub_flux", 14), ("sub_complexity", 14), ("sub_contrast", 14),
    ("spectral_contrast", 1), ("amplitude_peak", 1), ("amplitude_rms", 1), ("magnitude", 1024),
    ("amplitude_silence", 1), ("amplitude_envelope", 1), ("spectral_complexity", 1), ("auto_correlation", 1),
    ("f0", 1), ("f0_confidence", 1), ("failsafe_f0", 1), ("spectral_inharmonicity", 1),
    ("tristimulus1", 1), ("tristimulus2", 1), ("tristimulus3", 1),
]
FIELD_MASK = {
    "mfcc": D_MFCC, "spectral_rms": D_SPECTRAL_RMS, "spectral_centroid": D_SPECTRAL_CENTROID,
    "spectral_spread": D_SPECTRAL_SPREAD, "spectral_skewness": D_SPECTRAL_SKEWNESS,
    "spectral_kurtosis": D_SPECTRAL_KURTOSIS, "spectral_rolloff": D_SPECTRAL_ROLLOFF,
    "spectral_flatness": D_SPECTRAL_FLATNESS, "spectral_flux": D_SPECTRAL_FLUX,
    "spectrum_bands": D_SPECTRUM_BANDS, "sub_rms": D_BAND_FEATURES, "sub_flatness": D_BAND_FEATURES,
    "sub_flux": D_BAND_FEATURES, "sub_complexity": D_BAND_FEATURES, "sub_contrast": D_BAND_FEATURES,
    "spectral_contrast": D_BAND_FEATURES, "amplitude_peak": D_AMPLITUDE_PEAK,
    "amplitude_rms": D_AMPLITUDE_RMS, "magnitude": D_MAGNITUDE,
    "amplitude_silence": D_AMPLITUDE_SILENCE, "amplitude_envelope": D_AMPLITUDE_ENVELOPE,
    "spectral_complexity": D_SPECTRAL_COMPLEXITY, "auto_correlation": D_AUTO_CORRELATION,
    "f0": D_F0, "f0_confidence": D_F0, "failsafe_f0": D_F0,
    "spectral_inharmonicity": D_SPECTRAL_INHARMONICITY,
    "tristimulus1": D_TRISTIMULUS, "tristimulus2": D_TRISTIMULUS, "tristimulus3": D_TRISTIMULUS,
}


class AfxError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"afx status {status}: {text}")
        self.status = status


class _PlanDesc(ctypes.Structure):
    _fields_ = [("sample_rate", ctypes.c_int32), ("fft_size", ctypes.c_int32),
                ("hop_size", ctypes.c_int32), ("device", ctypes.c_int32),
                ("precision", ctypes.c_int32), ("max_analysis_ms", ctypes.c_int32),
                ("frame_kernel", ctypes.c_int32), ("flags", ctypes.c_int32)]


class _BatchInfo(ctypes.Structure):
    _fields_ = [("frame_kernel", ctypes.c_int32), ("feature_class", ctypes.c_int32), ("pcm_kind", ctypes.c_int32),
                ("chunk_frames", ctypes.c_int32), ("n_chunks", ctypes.c_int32), ("grid_blocks", ctypes.c_int32),
                ("arena_bytes", ctypes.c_int64)]


class _Buf(ctypes.Structure):
    _fields_ = [("pcm", ctypes.c_void_p), ("dtype", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("n_samples", ctypes.c_int64)]


class _Out(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n, _ in OUT_FIELDS] + [
        ("effective_length", ctypes.c_void_p), ("frame_offset", ctypes.c_void_p), ("buf_status", ctypes.c_void_p)]


class _Raw(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("format", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("sample_rate", ctypes.c_int32), ("reserved", ctypes.c_int32), ("n_frames", ctypes.c_int64)]


class _LoadInfo(ctypes.Structure):
    _fields_ = [("peak_value", ctypes.c_float), ("rms_value", ctypes.c_float), ("data_offset", ctypes.c_int32),
                ("silent_leading", ctypes.c_int32), ("silent_trailing", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("n_samples", ctypes.c_int64)]


class _StatsOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n, _ in OUT_FIELDS if n != "magnitude"] + [("stats_status", ctypes.c_void_p)]


def build_info():
    """afx_build_info() of the loaded library: "afx abi=N arch=gfx950 stamps=0 ablation=0 src=<hash of its sources>"."""
    return load_library().afx_build_info().decode()


def library_path():
    return _LIB_PATH


def device_count():
    """afx_device_count(): HIP devices this process can use (0 without a GPU)."""
    L = load_library()
    L.afx_device_count.restype = ctypes.c_int
    return int(L.afx_device_count())


def build_library(force=False):
    """Compile libafx_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src_dir]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH) and "AFX_LIBRARY" not in os.environ:
        try:
            build_library()      # hipcc is part of the image, on the GPU box as well
        except Exception as e:   # noqa: BLE001
            raise ImportError(f"{_LIB_PATH} is missing and could not be built ({e}); the HIP extension is "
                              "the only compute path, there is no fallback") from e
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing (the HIP extension is the only compute path; there is no fallback)")
    L = ctypes.CDLL(_LIB_PATH)
    vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
    L.afx_status_str.restype = ctypes.c_char_p
    L.afx_status_str.argtypes = [ctypes.c_int]
    L.afx_last_error.restype = ctypes.c_char_p
    L.afx_build_info.restype = ctypes.c_char_p
    L.afx_plan_create.argtypes = [ctypes.POINTER(_PlanDesc), ctypes.POINTER(vp)]
    L.afx_plan_destroy.argtypes = [vp]
    L.afx_plan_destroy.restype = None
    L.afx_plan_get_window.argtypes = [vp, vp]
    L.afx_plan_get_mel_table.argtypes = [vp, vp]
    L.afx_plan_get_bin_range.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.afx_num_frames.restype = i64
    L.afx_num_frames.argtypes = [vp, i64]
    L.afx_extract_batch.argtypes = [vp, ctypes.POINTER(_Buf), i32, u32, ctypes.POINTER(_Out)]
    L.afx_batch_create.argtypes = [vp, ctypes.POINTER(_Buf), i32, u32, ctypes.POINTER(vp)]
    L.afx_batch_total_frames.restype = i64
    L.afx_batch_total_frames.argtypes = [vp]
    L.afx_batch_run.argtypes = [vp]
    L.afx_batch_sync.argtypes = [vp]
    L.afx_batch_run_timed.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float)]
    L.afx_batch_fetch.argtypes = [vp, ctypes.POINTER(_Out)]
    L.afx_batch_fetch_statistics.argtypes = [vp, ctypes.POINTER(_StatsOut)]
    L.afx_batch_destroy.argtypes = [vp]
    L.afx_batch_destroy.restype = None
    L.afx_batch_create_from_raw.argtypes = [vp, ctypes.POINTER(_Raw), i32, u32, ctypes.POINTER(vp), ctypes.POINTER(_LoadInfo)]
    L.afx_batch_fetch_samples.argtypes = [vp, i32, vp, i64]
    L.afx_host_alloc.restype = vp
    L.afx_host_alloc.argtypes = [i64]
    L.afx_host_free.argtypes = [vp]
    L.afx_host_free.restype = None
    L.afx_batch_set_file_info.argtypes = [vp, vp]
    L.afx_batch_rhythm_frames.restype = i64
    L.afx_batch_rhythm_frames.argtypes = [vp, vp]
    L.afx_batch_fetch_rhythm.argtypes = [vp, vp, vp, vp]
    L.afx_batch_get_info.argtypes = [vp, ctypes.POINTER(_BatchInfo)]
    L.afx_batch_fetch_onset_functions.argtypes = [vp, vp]
    L.afx_algorithmic_bytes_per_frame.restype = i64
    L.afx_algorithmic_bytes_per_frame.argtypes = [vp, u32, i32]
    _lib = L
    return L


def pinned_array(shape, dtype):
    """numpy array in page-locked host memory (afx_host_alloc); keep the returned owner alive."""
    L = load_library()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = L.afx_host_alloc(max(n, 1))
    if not p:
        raise MemoryError("afx_host_alloc failed")

    class _Owner:
        def __del__(self, p=p, L=L):
            L.afx_host_free(p)

    buf = (ctypes.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    return arr, _Owner()


def _check(L, st):
    if st != 0:
        raise AfxError(st, (L.afx_status_str(st) or b"").decode() + ": " + (L.afx_last_error() or b"").decode())


def _pack_bufs(bufs):
    """list of 1-D float32/float64 arrays -> (ctypes array of afx_buf, keep-alive list)."""
    keep = []
    arr = (_Buf * max(1, len(bufs)))()
    for i, b in enumerate(bufs):
        b = np.asarray(b)
        if b.dtype == np.float32:
            dt = PCM_F32
        elif b.dtype == np.float64:
            dt = PCM_F64
        else:
            raise TypeError("PCM buffers must be float32 or float64")
        b = np.ascontiguousarray(b.reshape(-1))
        keep.append(b)
        arr[i].pcm = b.ctypes.data if b.size else None
        arr[i].dtype = dt
        arr[i].n_samples = b.size
    return arr, keep


def _alloc_out(mask, total_frames, n_bufs):
    out = _Out()
    res = {}
    for name, width in OUT_FIELDS:
        if mask & FIELD_MASK[name]:
            a = np.zeros((total_frames, width) if width > 1 else (total_frames,), dtype=np.float64)
            res[name] = a
            setattr(out, name, a.ctypes.data if a.size else None)
    if mask & D_EFFECTIVE_LENGTH:
        el = np.zeros((max(1, n_bufs), 3), dtype=np.float64)
        out.effective_length = el.ctypes.data
        res["effective_length"] = el[:n_bufs]
    fo = np.zeros(n_bufs + 1, dtype=np.int64)
    bs = np.zeros(max(1, n_bufs), dtype=np.int32)
    out.frame_offset = fo.ctypes.data
    out.buf_status = bs.ctypes.data
    res["frame_offset"] = fo
    res["buf_status"] = bs[:n_bufs]
    return out, res


class Plan:
    """afx_plan: the analogue of constructing TSampleAnalyser(44100, 2048, 1024)."""

    def __init__(self, sample_rate=44100, fft_size=2048, hop_size=1024, device=0,
                 precision=PRECISION_F64, max_analysis_ms=20000, frame_kernel=FRAME_KERNEL_AUTO, flags=0):
        self.L = load_library()
        d = _PlanDesc(sample_rate, fft_size, hop_size, device, precision, max_analysis_ms, frame_kernel, flags)
        h = ctypes.c_void_p()
        _check(self.L, self.L.afx_plan_create(ctypes.byref(d), ctypes.byref(h)))
        self.h = h
        self.fft_size, self.hop_size, self.precision = fft_size, hop_size, precision

    def close(self):
        if getattr(self, "h", None):
            self.L.afx_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def window(self):
        a = np.zeros(self.fft_size)
        _check(self.L, self.L.afx_plan_get_window(self.h, a.ctypes.data))
        return a

    def mel_table(self):
        a = np.zeros((14, self.fft_size // 2))
        _check(self.L, self.L.afx_plan_get_mel_table(self.h, a.ctypes.data))
        return a

    def bin_range(self):
        f, c = ctypes.c_int32(), ctypes.c_int32()
        _check(self.L, self.L.afx_plan_get_bin_range(self.h, ctypes.byref(f), ctypes.byref(c)))
        return f.value, c.value

    def num_frames(self, n_samples):
        return int(self.L.afx_num_frames(self.h, int(n_samples)))

    def bytes_per_frame(self, mask, pcm_dtype=PCM_F32):
        return int(self.L.afx_algorithmic_bytes_per_frame(self.h, mask, pcm_dtype))

    def extract(self, bufs, mask=D_ALL_LOW_LEVEL):
        """One-shot afx_extract_batch on host buffers; returns dict of numpy arrays."""
        arr, keep = _pack_bufs(bufs)
        total = sum(self.num_frames(b.size) for b in keep)
        out, res = _alloc_out(mask, total, len(keep))
        _check(self.L, self.L.afx_extract_batch(self.h, arr, len(keep), mask, ctypes.byref(out)))
        del keep
        return res

    def batch(self, bufs, mask=D_ALL_LOW_LEVEL):
        return Batch(self, bufs, mask)

    def batch_from_raw(self, raws, mask=D_ALL_LOW_LEVEL):
        """raws: list of (array, channels[, sample_rate]); array dtype int16 / int32 / float32 / float64 (interleaved,
        shape [frames*channels] or [frames, channels]) or uint8 of packed 24-bit little-endian samples; files at another
        sample_rate than the plan's are converted on the GPU (SampleAnalyser.cpp:563-607).
        Returns (Batch, list of load-info dicts): the LoadSample front end on the GPU."""
        n = len(raws)
        arr = (_Raw * max(1, n))()
        keep = []
        for i, item in enumerate(raws):
            data, channels = item[0], int(item[1])
            rate = int(item[2]) if len(item) > 2 else 0
            data = np.ascontiguousarray(np.asarray(data).reshape(-1))
            if data.dtype == np.int16:
                fmt, frames = RAW_I16, data.size // max(channels, 1)
            elif data.dtype == np.float32:
                fmt, frames = RAW_F32, data.size // max(channels, 1)
            elif data.dtype == np.uint8:
                fmt, frames = RAW_I24, data.size // (3 * max(channels, 1))
            elif data.dtype == np.int32:
                fmt, frames = RAW_I32, data.size // max(channels, 1)
            elif data.dtype == np.float64:
                fmt, frames = RAW_F64, data.size // max(channels, 1)
            else:
                raise TypeError("raw PCM must be int16, int32, float32, float64 or uint8 (packed int24)")
            keep.append(data)
            arr[i].data = data.ctypes.data if data.size else None
            arr[i].format, arr[i].channels, arr[i].sample_rate, arr[i].n_frames = fmt, channels, rate, frames
        info = (_LoadInfo * max(1, n))()
        h = ctypes.c_void_p()
        _check(self.L, self.L.afx_batch_create_from_raw(self.h, arr, n, mask, ctypes.byref(h), info))
        b = Batch.__new__(Batch)
        b.plan, b.L, b.mask, b.n_bufs, b.h = self, self.L, mask, n, h
        b.total_frames = int(self.L.afx_batch_total_frames(h))
        infos = [{k: getattr(info[i], k) for k, _ in _LoadInfo._fields_ if k != "reserved"} for i in range(n)]
        return b, infos


class Batch:
    """afx_batch: PCM resident in HBM, re-runnable."""

    def __init__(self, plan, bufs, mask):
        self.plan, self.L, self.mask = plan, plan.L, mask
        arr, keep = _pack_bufs(bufs)
        self.n_bufs = len(keep)
        h = ctypes.c_void_p()
        _check(self.L, self.L.afx_batch_create(plan.h, arr, self.n_bufs, mask, ctypes.byref(h)))
        self.h = h
        self.total_frames = int(self.L.afx_batch_total_frames(h))

    def run(self):
        _check(self.L, self.L.afx_batch_run(self.h))

    def info(self):
        """afx_batch_get_info: which STFT kernel the batch launches, its chunking, the PCM kept in HBM."""
        i = _BatchInfo()
        _check(self.L, self.L.afx_batch_get_info(self.h, ctypes.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in _BatchInfo._fields_}

    def sync(self):
        _check(self.L, self.L.afx_batch_sync(self.h))

    def run_timed(self, steps):
        """steps launches bracketed by HIP events on the batch stream -> elapsed ms."""
        ms = ctypes.c_float()
        _check(self.L, self.L.afx_batch_run_timed(self.h, steps, ctypes.byref(ms)))
        return ms.value

    def fetch(self):
        out, res = _alloc_out(self.mask, self.total_frames, self.n_bufs)
        _check(self.L, self.L.afx_batch_fetch(self.h, ctypes.byref(out)))
        return res

    def fetch_samples(self, buf, n):
        a = np.zeros(n, dtype=np.float64)
        _check(self.L, self.L.afx_batch_fetch_samples(self.h, buf, a.ctypes.data, n))
        return a

    def fetch_statistics(self):
        """dict name -> [n_bufs, W, 13] (W squeezed for scalar series) + "stats_status"."""
        out = _StatsOut()
        res = {}
        for name, width in OUT_FIELDS:
            if name == "magnitude" or not (self.mask & FIELD_MASK[name]):
                continue
            a = np.zeros((self.n_bufs, width, NUM_STATISTICS) if width > 1 else (self.n_bufs, NUM_STATISTICS))
            res[name] = a
            setattr(out, name, a.ctypes.data if a.size else None)
        st = np.zeros(max(1, self.n_bufs), dtype=np.int32)
        out.stats_status = st.ctypes.data
        _check(self.L, self.L.afx_batch_fetch_statistics(self.h, ctypes.byref(out)))
        res["stats_status"] = st[:self.n_bufs]
        return res

    def set_file_info(self, info):
        """info: per buffer (original_sample_rate, data_offset, original_samples) -- what TSampleData carries besides
        the samples; the rhythm tracker's duration heuristics use it (SampleAnalyser.cpp:1001-1004)."""
        a = np.zeros(max(1, self.n_bufs), dtype=[("rate", np.int32), ("offset", np.int32), ("samples", np.int64)])
        for i, (rate, offset, samples) in enumerate(info):
            a[i] = (rate, offset, samples)
        _check(self.L, self.L.afx_batch_set_file_info(self.h, a.ctypes.data))

    def rhythm_frames(self):
        off = np.zeros(self.n_bufs + 1, dtype=np.int64)
        self.L.afx_batch_rhythm_frames(self.h, off.ctypes.data)
        return off

    def fetch_rhythm(self, statistics=False, onset_functions=False):
        """dict: "offsets" [n_bufs+1], "onsets" [T][2] (complex, percussive), "scalars" [n_bufs][14] (RHYTHM_SCALARS
        order), optionally "onset_statistics" [n_bufs][2][13] and "onset_functions" float32 [T][2]."""
        off = self.rhythm_frames()
        rows = int(off[-1])
        res = {"offsets": off, "onsets": np.zeros((rows, 2)), "scalars": np.zeros((self.n_bufs, len(RHYTHM_SCALARS)))}
        st = np.zeros((self.n_bufs, 2, NUM_STATISTICS)) if statistics else None
        _check(self.L, self.L.afx_batch_fetch_rhythm(self.h, res["onsets"].ctypes.data if rows else None,
                                                     res["scalars"].ctypes.data if self.n_bufs else None,
                                                     st.ctypes.data if statistics and self.n_bufs else None))
        if statistics:
            res["onset_statistics"] = st
        if onset_functions:
            odf = np.zeros((max(rows, 1), 2), dtype=np.float32)
            _check(self.L, self.L.afx_batch_fetch_onset_functions(self.h, odf.ctypes.data))
            res["onset_functions"] = odf[:rows]
        return res

    def close(self):
        if getattr(self, "h", None):
            self.L.afx_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
