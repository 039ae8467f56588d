/* include/afx.h -- C-ABI of the MI355X low-level audio feature extractor (libafx_hip.so).
 *
 * This is the drop-in boundary for the per-frame spectral loop of AFEC's
 * TSampleAnalyser::AnalyzeLowLevelDescriptors
 * (Source/Crawler/FeatureExtraction/Source/SampleAnalyser.cpp:814-976).  The reference has no FFI
 * for this path (the loop body is private member code, Export/SampleAnalyser.h:65-211); the entry
 * points below are what a maintainer binds from that function -- see INTEGRATION.md for the
 * reference-side stub.  Plain C types only: no torch, no C++ types, no exceptions cross this ABI.
 *
 * Every entry point runs on the GPU (HIP, gfx950).  There is no CPU fallback: if no HIP device is
 * present afx_plan_create fails with AFX_ERR_NO_DEVICE.
 */
#ifndef AFX_H
#define AFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFX_VERSION 6

/* ---- status codes (replace TReadableException on this path, SampleAnalyser.cpp:397-408) ---- */
enum {
  AFX_OK = 0,
  AFX_ERR_INVALID_ARG = -1,
  AFX_ERR_UNSUPPORTED = -2, /* plan geometry the HIP kernels are not specialised for; a file's rate */
  AFX_ERR_NO_DEVICE = -3,   /* no HIP device / device ordinal out of range                  */
  AFX_ERR_OUT_OF_MEMORY = -4,
  AFX_ERR_HIP = -5,         /* a HIP runtime call failed; afx_last_error() has the text      */
  AFX_ERR_BAD_BUFFER = -6   /* per-buffer status: NULL pcm / negative length / unknown dtype */
};
const char* afx_status_str(int status);
const char* afx_last_error(void); /* thread-local detail text of the last failing call */
/* "afx abi=N arch=gfx950 stamps=0 ablation=0 src=<hash of the library's sources>": the shipped library carries no
 * diagnostic / ablation switch; the hash ties committed profiler counters to the build they were measured on */
const char* afx_build_info(void);

/* ---- PCM sample types of afx_buf.dtype ---- */
enum {
  AFX_PCM_F32 = 0, /* float, nominal range [-1, 1]  */
  AFX_PCM_F64 = 1  /* double, as TSampleData::mData (Export/SampleAnalyser.h:86-88) */
};

/* ---- internal arithmetic of the STFT ---- */
enum {
  AFX_PRECISION_F64 = 0, /* default: IEEE double end to end, like the reference             */
  AFX_PRECISION_F32 = 1  /* REMOVED (afx_plan_create answers AFX_ERR_UNSUPPORTED): float butterflies missed the
                            1e-4 bar on tonal input and were no faster than the double half-wave kernel */
};

/* ---- descriptor selection (bit mask) ---- *
 * Names follow TSampleDescriptors (Export/SampleDescriptors.h:407-465).                      */
enum {
  AFX_D_MFCC = 1u << 0,              /* cepstrum_bands      [F][14]  SampleAnalyser.cpp:2052-2063 */
  AFX_D_SPECTRAL_RMS = 1u << 1,      /* spectral_rms        [F]      :1808-1818 */
  AFX_D_SPECTRAL_CENTROID = 1u << 2, /* spectral_centroid   [F]      :1822-1837 */
  AFX_D_SPECTRAL_SPREAD = 1u << 3,   /* spectral_spread     [F]      :1822-1837 */
  AFX_D_SPECTRAL_SKEWNESS = 1u << 4, /* spectral_skewness   [F]      :1858-1883 */
  AFX_D_SPECTRAL_KURTOSIS = 1u << 5, /* spectral_kurtosis   [F]      :1858-1883 */
  AFX_D_SPECTRAL_ROLLOFF = 1u << 6,  /* spectral_rolloff    [F]      :1887-1901 */
  AFX_D_SPECTRAL_FLATNESS = 1u << 7, /* spectral_flatness   [F]      :1905-1915 */
  AFX_D_SPECTRAL_FLUX = 1u << 8,     /* spectral_flux       [F]      :1919-1933 */
  AFX_D_SPECTRUM_BANDS = 1u << 9,    /* frequency_bands     [F][28]  :2007-2048 */
  AFX_D_BAND_FEATURES = 1u << 10,    /* spectral_{rms,flatness,flux,complexity,contrast}_bands
                                        [F][14] each + spectral_contrast [F]  :2067-2308 */
  AFX_D_AMPLITUDE_PEAK = 1u << 11,   /* amplitude_peak      [F]      :1760-1770 */
  AFX_D_AMPLITUDE_RMS = 1u << 12,    /* amplitude_rms       [F]      :1774-1783 */
  AFX_D_MAGNITUDE = 1u << 13,        /* magnitude spectrum  [F][1024] for CPU-resident consumers
                                        (whitening, pitch; SampleAnalyser.cpp:850-927)        */
  AFX_D_STATISTICS = 1u << 14,       /* additionally reduce every selected series of every buffer to
                                        the 13 values of TStatistics::Calc (Statistics.cpp:12-90),
                                        as TSampleAnalyser::CalcStatistics does (SampleAnalyser.cpp:2402) */
  /* the loop's neighbours of the spectral set: time-domain and whitened-spectrum
   * descriptors of the same frames, SampleAnalyser.cpp:849-927, 942-964 */
  AFX_D_AMPLITUDE_SILENCE = 1u << 15,    /* amplitude_silence   [F]  :865-868 (aubio_silence_detection, -48 dB) */
  AFX_D_AMPLITUDE_ENVELOPE = 1u << 16,   /* amplitude_envelope  [F]  :1787-1804 (TEnvelopeDetector kFast, 8 ms) */
  AFX_D_SPECTRAL_COMPLEXITY = 1u << 17,  /* spectral_complexity [F]  :849-862, 1937-1947 (adaptive whitening,
                                            peak spectrum, peaks inside the analysis range)  */
  AFX_D_AUTO_CORRELATION = 1u << 18,     /* auto_correlation    [F]  :2312-2398 */
  AFX_D_F0 = 1u << 19,                   /* f0, f0_confidence, failsafe_f0 [F] each  :876-917 (aubio yinfast) */
  AFX_D_SPECTRAL_INHARMONICITY = 1u << 20, /* spectral_inharmonicity [F]  :1951-1971 */
  AFX_D_TRISTIMULUS = 1u << 21,          /* tristimulus1..3     [F] each  :1975-2003 */
  /* per file, not per frame (SampleAnalyser.cpp:754, 1715-1755) */
  AFX_D_EFFECTIVE_LENGTH = 1u << 22,     /* effectve_length_{48,24,12}dB  [n_bufs][3] seconds between the first and the
                                            last sample above -48 / -24 / -12 dB of the whole buffer (not only its
                                            analysed first 20 s) */
  /* per file, from a second STFT of its own (512-sample frames every 128 samples): the rhythm tracker,
   * SampleAnalyser.cpp:983-1048 (TRhythmTracker, TOnsetDetector, aubio beat tracking).  Results are fetched with
   * afx_batch_fetch_rhythm, not through afx_out. */
  AFX_D_RHYTHM = 1u << 23,
  AFX_D_C2 = AFX_D_MFCC,
  AFX_D_SPECTRAL_STATS = 0x1FEu,     /* bits 1..8 */
  AFX_D_ALL_LOW_LEVEL = 0x1FFFu,     /* the spectral set of SURVEY 8(a), everything except the raw magnitudes */
  AFX_D_NEIGHBOURS = 0x3F8000u,      /* bits 15..21 */
  AFX_D_ALL_PER_FRAME = 0x3F9FFFu    /* AFX_D_ALL_LOW_LEVEL | AFX_D_NEIGHBOURS */
};
#define AFX_NUM_CEPSTRUM 14 /* kNumberOfCepstrumCoefficients, SampleDescriptors.h:464 */
#define AFX_NUM_BANDS 28    /* kNumberOfSpectrumBands                                  */
#define AFX_NUM_SUBBANDS 14 /* kNumberOfSpectrumSubBands                               */

/* ---- plan: immutable analyser state, the analogue of the TSampleAnalyser ctor ---- *
 * (SampleAnalyser.cpp:162-198: analysis bin range, 2x Hann window, LibXtract mel table).
 * Safe to share between threads; every batch carries its own stream and workspace.        */
typedef struct afx_plan afx_plan;

typedef struct {
  int32_t sample_rate;     /* 44100 (Crawler.cpp:41)                                          */
  int32_t fft_size;        /* 2048  (Crawler.cpp:42)                                          */
  int32_t hop_size;        /* 1024  (Crawler.cpp:43)                                          */
  int32_t device;          /* HIP device ordinal                                              */
  int32_t precision;       /* AFX_PRECISION_*                                                 */
  int32_t max_analysis_ms; /* MAnalyzationDurationMaxInMs = 20000 (SampleAnalyser.cpp:37);
                              0 disables the cap (synthetic benchmarks)                       */
  int32_t frame_kernel;    /* AFX_FRAME_KERNEL_*: layout of the STFT kernel for the batches the half-wave layout serves
                              (float PCM); results agree to rounding (1e-6; discrete descriptors may flip), so a caller
                              whose results must not depend on how its files were batched pins WAVE64 or HALFWAVE
                              (the host layer's TSampleAnalyser / TCrawler do)                                    */
  int32_t flags;           /* AFX_PLAN_* bits                                                  */
} afx_plan_desc;
enum {
  AFX_FRAME_KERNEL_AUTO = 0,     /* by batch size (what a zeroed field means): fastest per batch, the layout -- and with it
                                    the last bits of a file's descriptors -- depends on the batch                  */
  AFX_FRAME_KERNEL_WAVE64 = 1,   /* one frame per 64-lane wave for every batch                                    */
  AFX_FRAME_KERNEL_HALFWAVE = 2  /* one frame per 32-lane half for every batch the layout serves                  */
};
enum {
  AFX_PLAN_NO_SIDE_STREAM = 1u << 0 /* the rhythm tracker's kernels run on the batch's own stream instead of beside the
                                       per-frame kernels (profiles: a kernel's duration is then its own; results are
                                       identical)                                                               */
};

/* Nothing in the environment changes what afx_plan_create builds or what a batch computes (ABI 4 read AFX_HALFWAVE and
 * AFX_SIDE_STREAM; they are afx_plan_desc.frame_kernel / flags now).
 * AFX_TIMING (any value, read when the library is loaded): wall time of the phases of afx_batch_create_from_raw, summed
 * over calls, printed to stderr when the process ends -- a diagnostic for pipelines, no effect on results.
 * The library does not touch the process' environment.  A pipeline that keeps several batches in flight per device
 * wants GPU_MAX_HW_QUEUES=16 (the HIP runtime's own variable: hardware queues its streams are multiplexed onto, 4 by
 * default; read when the runtime initialises): set it before the process' first HIP call, as afec::TCrawler does
 * (TCrawlOptions::mHardwareQueues).  More than the device's hardware queue slots (about 20) would be time-sliced. */
int afx_plan_create(const afx_plan_desc* desc, afx_plan** out_plan);
void afx_plan_destroy(afx_plan* plan);

/* introspection used by the parity tests (tables must equal LibXtract's bit for bit) */
int afx_plan_get_window(const afx_plan* plan, double* out /* [fft_size] */);
int afx_plan_get_mel_table(const afx_plan* plan, double* out /* [14][fft_size/2] */);
int afx_plan_get_bin_range(const afx_plan* plan, int32_t* first_bin, int32_t* bin_count);

/* frames the reference loop produces for n_samples (SampleAnalyser.cpp:760-764, 814) */
int64_t afx_num_frames(const afx_plan* plan, int64_t n_samples);

/* ---- input / output ---- */
typedef struct {
  const void* pcm;   /* host pointer, mono, AFX_PCM_* */
  int32_t dtype;
  int32_t reserved;
  int64_t n_samples;
} afx_buf;

/* Caller-allocated host arrays of doubles, row-major by frame over the whole batch (frames of
 * buffer i occupy rows frame_offset[i] .. frame_offset[i+1]-1), the layout of
 * TFramedScalarData::mValues / TFramedVectorData<W>::mValues (SampleDescriptors.h:152-356).
 * Any pointer may be NULL (not wanted); non-NULL pointers must be selected in the mask.     */
typedef struct {
  double* mfcc;              /* [F][14]   */
  double* spectral_rms;      /* [F]       */
  double* spectral_centroid; /* [F]       */
  double* spectral_spread;
  double* spectral_skewness;
  double* spectral_kurtosis;
  double* spectral_rolloff;
  double* spectral_flatness;
  double* spectral_flux;
  double* spectrum_bands;    /* [F][28]   */
  double* sub_rms;           /* [F][14]   */
  double* sub_flatness;      /* [F][14]   */
  double* sub_flux;          /* [F][14]   */
  double* sub_complexity;    /* [F][14]   */
  double* sub_contrast;      /* [F][14]   */
  double* spectral_contrast; /* [F]       */
  double* amplitude_peak;    /* [F]       */
  double* amplitude_rms;     /* [F]       */
  double* magnitude;         /* [F][1024] */
  double* amplitude_silence; /* [F], 1.0 = silent hop */
  double* amplitude_envelope;
  double* spectral_complexity;
  double* auto_correlation;
  double* f0;                /* Hz, 0 when the frame is silent or no period was found */
  double* f0_confidence;     /* 0..1 */
  double* failsafe_f0;
  double* spectral_inharmonicity;
  double* tristimulus1;
  double* tristimulus2;
  double* tristimulus3;
  double* effective_length;  /* [n_bufs][3]: -48, -24, -12 dB (per buffer; AFX_D_EFFECTIVE_LENGTH) */
  int64_t* frame_offset;     /* [n_bufs+1], optional */
  int32_t* buf_status;       /* [n_bufs], optional: AFX_OK or AFX_ERR_BAD_BUFFER (one bad buffer
                                does not fail the batch, cf. SampleAnalyser.cpp:368-408)      */
} afx_out;

/* Per-buffer statistics (AFX_D_STATISTICS): the same fields as afx_out, each an array
 * [n_bufs][W][AFX_NUM_STATISTICS] of doubles (W = 1 for scalar series), in the order of
 * TFramedScalarData's members (SampleDescriptors.h:172-186).  Series of any length are reduced (those of more
 * than 1024 frames, possible only with the 20 s cap disabled, by a streaming kernel); stats_status repeats
 * buf_status. */
#define AFX_NUM_STATISTICS 13
enum {
  AFX_S_MIN = 0, AFX_S_MAX, AFX_S_MEDIAN, AFX_S_MEAN, AFX_S_GMEAN, AFX_S_VARIANCE, AFX_S_CENTROID,
  AFX_S_SPREAD, AFX_S_SKEWNESS, AFX_S_KURTOSIS, AFX_S_FLATNESS, AFX_S_DMEAN, AFX_S_DVARIANCE
};
typedef struct {
  double* mfcc;              /* [n_bufs][14][13] */
  double* spectral_rms;      /* [n_bufs][13]     */
  double* spectral_centroid;
  double* spectral_spread;
  double* spectral_skewness;
  double* spectral_kurtosis;
  double* spectral_rolloff;
  double* spectral_flatness;
  double* spectral_flux;
  double* spectrum_bands;    /* [n_bufs][28][13] */
  double* sub_rms;           /* [n_bufs][14][13] */
  double* sub_flatness;
  double* sub_flux;
  double* sub_complexity;
  double* sub_contrast;
  double* spectral_contrast; /* [n_bufs][13]     */
  double* amplitude_peak;
  double* amplitude_rms;
  double* amplitude_silence;
  double* amplitude_envelope;
  double* spectral_complexity;
  double* auto_correlation;
  double* f0;
  double* f0_confidence;
  double* failsafe_f0;
  double* spectral_inharmonicity;
  double* tristimulus1;
  double* tristimulus2;
  double* tristimulus3;
  int32_t* stats_status;     /* [n_bufs], optional */
} afx_stats_out;

/* One-shot: upload n_bufs host buffers, run the HIP path, download the selected descriptors.
 * This is the call TSampleAnalyser::AnalyzeLowLevelDescriptors would make (n_bufs = 1 per file,
 * or many files per call from a batching crawler).  Thread-safe on a shared plan.            */
int afx_extract_batch(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask,
                      afx_out* out);

/* ---- resident batches: PCM stays in HBM across runs (pipelines, benchmarks) ---- */
typedef struct afx_batch afx_batch;

int afx_batch_create(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask,
                     afx_batch** out_batch);            /* allocates + uploads, synchronous   */
int64_t afx_batch_total_frames(const afx_batch* batch);
int afx_batch_run(afx_batch* batch);                    /* enqueue one pass on the batch stream */
int afx_batch_sync(afx_batch* batch);                   /* wait for the batch stream            */
/* enqueue `steps` passes bracketed by HIP events on the batch stream; returns the elapsed
 * device time of the bracket in milliseconds (the stream is idle on return)                   */
int afx_batch_run_timed(afx_batch* batch, int32_t steps, float* elapsed_ms);
int afx_batch_fetch(afx_batch* batch, afx_out* out);    /* D2H + unpack, synchronous            */
int afx_batch_fetch_statistics(afx_batch* batch, afx_stats_out* out); /* needs AFX_D_STATISTICS in the mask */
void afx_batch_destroy(afx_batch* batch);

/* What a batch will launch (decided when it is created): for tests and profiles that must know which kernel a result or
 * a duration belongs to. */
typedef struct {
  int32_t frame_kernel;  /* AFX_FRAME_KERNEL_WAVE64 or AFX_FRAME_KERNEL_HALFWAVE */
  int32_t feature_class; /* of the STFT kernel: 0 = MFCC only, 1 = + spectral statistics, 2 = + spectrum bands /
                            amplitude / stored magnitudes ("full"); half-wave kernel only: 3 = full with the whole
                            spectrum stored, 4 = MFCC + the whole spectrum stored (statistics from the band kernel)  */
  int32_t pcm_kind;      /* 0 = float, 1 = double, 2 = float + one double scale per buffer (LoadSample front end) */
  int32_t chunk_frames;  /* frames per chunk (the unit of work of a wave / half-wave)                          */
  int32_t n_chunks;
  int32_t grid_blocks;   /* workgroups of the STFT kernel                                                     */
  int64_t arena_bytes;   /* PCM kept in HBM for this batch                                                    */
} afx_batch_info;
int afx_batch_get_info(const afx_batch* batch, afx_batch_info* info);

/* Raw results for pipelines that keep many batches in flight: the per-frame records exactly as the kernels leave
 * them, double[total_frames][stride], and the statistics double[n_bufs][stride][AFX_NUM_STATISTICS], each moved by
 * ONE device-to-host transfer into caller memory (page-locked memory from afx_host_alloc makes it a direct DMA; no
 * intermediate copy, no per-field unpack).  offsets[i] is the first record column of the i-th series, -1 when the
 * series is not in the batch mask; widths[i] its number of columns.  Series order (= record order): mfcc,
 * spectral_rms, spectral_centroid, spectral_spread, spectral_skewness, spectral_kurtosis, spectral_rolloff,
 * spectral_flatness, spectral_flux, spectrum_bands, amplitude_peak, amplitude_rms, sub_rms, sub_flatness, sub_flux,
 * sub_complexity, sub_contrast, spectral_contrast, amplitude_silence, amplitude_envelope, spectral_complexity,
 * auto_correlation, f0, f0_confidence, failsafe_f0, spectral_inharmonicity, tristimulus1, tristimulus2, tristimulus3.
 * Either destination may be NULL. */
#define AFX_NUM_SERIES 29
int afx_batch_record_layout(const afx_batch* batch, int32_t* stride, int32_t* offsets /* [AFX_NUM_SERIES] */,
                            int32_t* widths /* [AFX_NUM_SERIES] */);
int afx_batch_fetch_records(afx_batch* batch, double* records, double* statistics, int64_t* frame_offset /* [n_bufs+1] */,
                            int32_t* buf_status /* [n_bufs] */, double* effective_length /* [n_bufs][3] or NULL */);

/* ---- LoadSample front end: TSampleAnalyser::LoadSample, SampleAnalyser.cpp:484-718 ---- *
 * Decoded, interleaved PCM of a file in; on the GPU: conversion to the reference's "16-bit float"
 * range, mono mix-down, peak / rms, peak normalisation, -48 dB leading / trailing silence trim, and the
 * half-frame / one-frame zero padding.  The result is the double buffer TSampleData::mData holds and
 * is kept in HBM as the float mono signal LoadSample itself works on plus the file's FinalScaling: mData[n] is their
 * product (SampleAnalyser.cpp:710-718) and every kernel forms it as it loads a sample (bit-identical, half the bytes);
 * afx_batch_fetch_samples returns the doubles.  Files that are not at the plan's rate (afx_raw.sample_rate) are converted
 * first, on the GPU, exactly as the reference does it on the CPU (SampleAnalyser.cpp:563-607: libresample 0.1.3,
 * resample_open(1, f, f) + one resample_process call, f = plan rate / file rate, on the mono mix; afx_resample.hip);
 * afx_load_info and the rhythm tracker's duration heuristics then see the file's own rate and length
 * (TSampleData::mOriginalSampleRate / mOriginalNumberOfSamples).  A file above 16 x the plan's rate gets
 * AFX_ERR_UNSUPPORTED in its buf_status, and so does a file whose conversion yields 2^30 samples or more (6.8 hours at
 * 44.1 kHz; the kernels index a file's samples with 32 bits): a header claiming 1 Hz must fail that file, not exhaust the
 * device for the batch.  (Until round 5: a rate below the plan's / 64, or 2^28 samples -- a two-hour recording at 48 kHz
 * failed.)  Decoding the container stays with the caller. */
enum {
  AFX_RAW_I16 = 0, /* int16                      (S16BitSignedTo16BitFloat, SampleConverter.h:446-449) */
  AFX_RAW_I24 = 1, /* packed little-endian int24 (S24BitTo16BitFloat, SampleConverter.h:474-486)       */
  AFX_RAW_F32 = 2, /* float in [-1, 1]           (S0To1FloatTo16BitFloat, SampleConverter.h:529-533)   */
  AFX_RAW_I32 = 3, /* int32                      (S32BitSignedTo16BitFloat, SampleConverter.h:514-518) */
  AFX_RAW_F64 = 4  /* double in [-1, 1]          (S0To1FloatTo16BitFloat, SampleConverter.h:529-533)   */
};
typedef struct {
  const void* data;    /* host pointer, interleaved by channel */
  int32_t format;      /* AFX_RAW_* */
  int32_t channels;    /* 1..8 (SampleAnalyser.cpp:472-477) */
  int32_t sample_rate; /* 0 = the plan's rate; another rate: converted on the GPU */
  int32_t reserved;
  int64_t n_frames;    /* sample frames per channel */
} afx_raw;
typedef struct {
  float peak_value;        /* TSampleData::mPeakValue  */
  float rms_value;         /* TSampleData::mRmsValue   */
  int32_t data_offset;     /* TSampleData::mDataOffset */
  int32_t silent_leading;
  int32_t silent_trailing;
  int32_t reserved;
  int64_t n_samples;       /* TSampleData::mData.Size() */
} afx_load_info;
int afx_batch_create_from_raw(afx_plan* plan, const afx_raw* raws, int32_t n_bufs, uint32_t mask,
                              afx_batch** out_batch, afx_load_info* info /* [n_bufs], optional */);
/* the normalised samples of buffer `buf` (its analysed prefix) for the CPU-resident neighbours */
int afx_batch_fetch_samples(afx_batch* batch, int32_t buf, double* dst, int64_t n);

/* ---- rhythm tracker (AFX_D_RHYTHM): SampleAnalyser.cpp:983-1048 ---- *
 * Per file: the two onset series of TRhythmTracker::Onsets (one value per 512/128 frame of the analysed prefix: the
 * median-removed onset function where an onset was detected, 0 elsewhere), and 14 scalars.  Everything runs on the GPU.
 * The final tempo's duration heuristics need what TSampleData carries besides the samples (SampleAnalyser.cpp:
 * 1001-1004): batches made by afx_batch_create_from_raw know it (the file's frames, the plan's rate, the data offset
 * LoadSample produced); for afx_batch_create the defaults are the buffer's own length, the plan's rate and offset 0 --
 * afx_batch_set_file_info overrides them (call it before afx_batch_run). */
#define AFX_NUM_RHYTHM_SCALARS 14
enum {
  AFX_R_COMPLEX_ONSET_COUNT = 0,      /* rhythm_complex_onset_count            RhythmTracker.cpp:124-137 */
  AFX_R_COMPLEX_TEMPO,                /* rhythm_complex_tempo                  :159-234 (aubio beattracking.c) */
  AFX_R_COMPLEX_TEMPO_CONFIDENCE,     /* rhythm_complex_tempo_confidence                                */
  AFX_R_COMPLEX_ONSET_FREQUENCY_MEAN, /* rhythm_complex_onset_frequency_mean   :270-296 */
  AFX_R_COMPLEX_ONSET_STRENGTH,       /* rhythm_complex_onset_strength         :300-325 */
  AFX_R_COMPLEX_ONSET_CONTRAST,       /* rhythm_complex_onset_contrast         :329-412 */
  AFX_R_PERCUSSIVE_ONSET_COUNT,       /* the same six for the percussive (power) onset function */
  AFX_R_PERCUSSIVE_TEMPO,
  AFX_R_PERCUSSIVE_TEMPO_CONFIDENCE,
  AFX_R_PERCUSSIVE_ONSET_FREQUENCY_MEAN,
  AFX_R_PERCUSSIVE_ONSET_STRENGTH,
  AFX_R_PERCUSSIVE_ONSET_CONTRAST,
  AFX_R_FINAL_TEMPO,                  /* rhythm_final_tempo             SampleAnalyser.cpp:1029-1048, RhythmTracker.cpp:238-325 */
  AFX_R_FINAL_TEMPO_CONFIDENCE        /* rhythm_final_tempo_confidence */
};
typedef struct {
  int32_t original_sample_rate; /* TSampleData::mOriginalSampleRate      */
  int32_t data_offset;          /* TSampleData::mDataOffset              */
  int64_t original_samples;     /* TSampleData::mOriginalNumberOfSamples */
} afx_file_info;
int afx_batch_set_file_info(afx_batch* batch, const afx_file_info* info /* [n_bufs] */);
/* 512/128 frames of every buffer: offsets[i] .. offsets[i+1]-1 are buffer i's rows of `onsets`; returns the total */
int64_t afx_batch_rhythm_frames(const afx_batch* batch, int64_t* offsets /* [n_bufs+1] or NULL */);
/* onsets [total][2] (complex, percussive), scalars [n_bufs][AFX_NUM_RHYTHM_SCALARS], onset_statistics
 * [n_bufs][2][AFX_NUM_STATISTICS] (needs AFX_D_STATISTICS: the reference's CalcStatistics covers the two onset
 * series).  Any destination may be NULL. */
int afx_batch_fetch_rhythm(afx_batch* batch, double* onsets, double* scalars, double* onset_statistics);
/* the onset functions before median removal, float [total][2]: for the parity tests */
int afx_batch_fetch_onset_functions(afx_batch* batch, float* odf);

/* Page-locked host memory for PCM and result arrays: transfers from / to such buffers run at the
 * host link's rate (pageable memory is staged by the runtime at a fraction of it). */
void* afx_host_alloc(int64_t bytes);
void afx_host_free(void* p);

/* How host threads wait for the device in afx_batch_create_from_raw, afx_batch_fetch_records and afx_batch_fetch_rhythm
 * of this plan's batches: spinning (the HIP runtime's default: lowest latency, one busy CPU per waiting thread) or, with
 * blocking != 0, sleeping between polls of an event (20 us naps).  A pipeline with several batches in flight per device
 * wants the latter: the streaming driver of afec_amd/host keeps eight worker threads per GPU and reaches the same
 * 270 k files/s with about 3 instead of 7 busy CPUs (TCrawlOptions::mSleepingWaits).  Per plan; applies to the batches
 * created after the call.  (ABI 3 had a process-wide afx_set_blocking_wait instead.) */
int afx_plan_set_blocking_wait(afx_plan* plan, int32_t blocking);

/* HIP devices this process can use (0 when there is none or the runtime cannot be reached): the ordinals a host layer
 * shards files over, file i -> device i mod G (Crawler.cpp:706-728 hands one self-contained task per file to a pool of
 * threads; afec::TCrawlOptions::mDevices).  ABI 6. */
int afx_device_count(void);

/* Does the plan's device still answer?  AFX_OK, or AFX_ERR_HIP when the runtime reports an error for a trivial request
 * on this device (after a fault that took the context down every call fails).  A caller whose batch failed uses this to
 * tell "this batch cannot be analysed" (retry smaller, record the file as failed, go on: SampleAnalyser.cpp:368-408)
 * from "nothing more can be analysed".  Cheap: allocates nothing (after an out-of-memory failure an allocation could
 * fail on a live device) and waits for nothing but a 4-byte write on a stream of its own. */
int afx_plan_probe_device(afx_plan* plan);

/* static facts for roofline accounting (bytes the algorithm must move per frame for `mask`) */
int64_t afx_algorithmic_bytes_per_frame(const afx_plan* plan, uint32_t mask, int32_t pcm_dtype);

#ifdef __cplusplus
}
#endif
#endif /* AFX_H */
