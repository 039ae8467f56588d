"""afec_amd -- MI355X-native low-level audio feature extraction (the AFEC crawler's per-frame
spectral hot path as HIP kernels behind the C-ABI of include/afx.h).

Python here is test/bench plumbing over the C-ABI (ctypes); the product is
afec_amd/lib/libafx_hip.so and the C++ host layer in afec_amd/host/.
"""
from .capi import (  # noqa: F401
    AfxError, Batch, Plan, build_info, build_library, device_count, library_path, load_library,
    D_MFCC, D_SPECTRAL_RMS, D_SPECTRAL_CENTROID, D_SPECTRAL_SPREAD, D_SPECTRAL_SKEWNESS,
    D_SPECTRAL_KURTOSIS, D_SPECTRAL_ROLLOFF, D_SPECTRAL_FLATNESS, D_SPECTRAL_FLUX,
    D_SPECTRUM_BANDS, D_BAND_FEATURES, D_AMPLITUDE_PEAK, D_AMPLITUDE_RMS, D_MAGNITUDE, D_STATISTICS, NUM_STATISTICS, STAT_NAMES,
    D_AMPLITUDE_SILENCE, D_AMPLITUDE_ENVELOPE, D_SPECTRAL_COMPLEXITY, D_AUTO_CORRELATION, D_F0,
    D_SPECTRAL_INHARMONICITY, D_TRISTIMULUS, D_EFFECTIVE_LENGTH, D_RHYTHM, RHYTHM_SCALARS, D_NEIGHBOURS, D_ALL_PER_FRAME,
    D_C2, D_SPECTRAL_STATS, D_ALL_LOW_LEVEL, PRECISION_F64, PRECISION_F32, PCM_F32, PCM_F64,
    FRAME_KERNEL_AUTO, FRAME_KERNEL_WAVE64, FRAME_KERNEL_HALFWAVE, PLAN_NO_SIDE_STREAM,
)
