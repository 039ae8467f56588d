/* oracle/afx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the AFEC low-level hot path, used solely as
 * the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product (afec_amd/, libafx_hip.so) never includes, links or calls anything in oracle/.
 * Parity status: PINNED -- checked against the reference's own compiled objects
 * (oracle/_ref/ref_driver, see oracle/Makefile) and against tests/golden/ fixtures generated
 * from them (tests/golden/make_golden.py).
 */
#ifndef AFX_ORACLE_H
#define AFX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* per-frame record layout in doubles (identical to oracle/ref_driver.cpp) */
enum {
  AFXO_MAG = 0, AFXO_MFCC = 1024, AFXO_SRMS = 1038, AFXO_CENTROID, AFXO_SPREAD, AFXO_SKEW,
  AFXO_KURT, AFXO_ROLLOFF, AFXO_FLATNESS, AFXO_FLUX, AFXO_BANDS = 1046, AFXO_SUB_RMS = 1074,
  AFXO_SUB_FLAT = 1088, AFXO_SUB_FLUX = 1102, AFXO_SUB_CPLX = 1116, AFXO_SUB_CONTRAST = 1130,
  AFXO_CONTRAST = 1144, AFXO_AMP_PEAK, AFXO_AMP_RMS, AFXO_RECORD
};

typedef struct afx_oracle afx_oracle;

afx_oracle* afx_oracle_create(int sample_rate, int fft_size, int hop_size);
void afx_oracle_destroy(afx_oracle*);
const double* afx_oracle_window(const afx_oracle*);        /* [fft_size]            */
const double* afx_oracle_mel(const afx_oracle*);           /* [14][fft_size/2]      */
int afx_oracle_first_bin(const afx_oracle*);
int afx_oracle_bin_count(const afx_oracle*);

/* frames the reference loop produces for a buffer of n_samples (SampleAnalyser.cpp:760-764, 814) */
int64_t afx_oracle_num_frames(const afx_oracle*, int64_t n_samples, int apply_cap);

/* run the per-frame loop over one buffer; records = [frames][AFXO_RECORD]; returns frames */
int64_t afx_oracle_run(const afx_oracle*, const double* x, int64_t n_samples, int apply_cap,
                       double* records);

/* C2 subset only (window -> FFT -> magnitude -> MFCC), for the bench cpu_baseline leg:
 * mfcc = [frames][14]; returns frames */
int64_t afx_oracle_run_mfcc(const afx_oracle*, const double* x, int64_t n_samples, double* mfcc);

/* ---- stateful neighbours of the loop (SampleAnalyser.cpp:849-927, 942-964) ----
 * PINNED against oracle/_ref/ref_driver `neighbours`, which runs the reference's own aubio
 * (whitening, silence, yinfast pitch), TEnvelopeDetector, TAutocorrelation and LibXtract objects.
 * record = silence, envelope, f0, f0 confidence, fail-safe f0, autocorrelation, spectral complexity,
 * inharmonicity, tristimulus 1..3, whitened spectrum [fft/2]. */
enum {
  AFXN_SILENCE = 0, AFXN_ENVELOPE, AFXN_F0, AFXN_F0_CONF, AFXN_F0_FAILSAFE, AFXN_AUTOCORR,
  AFXN_COMPLEXITY, AFXN_INHARM, AFXN_TRI1, AFXN_TRI2, AFXN_TRI3, AFXN_WHITE, AFXN_RECORD = AFXN_WHITE + 1024
};
int64_t afx_oracle_run_neighbours(const afx_oracle*, const double* x, int64_t n_samples, int apply_cap,
                                  double* records);
/* TStatistics::Peaks (Statistics.cpp:140-232): bins/vals sized n; returns the number of peaks */
int afx_oracle_peaks(const double* x, int n, double threshold, int* bins, double* vals);

/* CalcEffectiveLength (SampleAnalyser.cpp:1715-1755): seconds between the first and last sample above
 * -48 / -24 / -12 dB of the whole buffer.  PINNED against oracle/_ref/ref_driver `efflen`. */
void afx_oracle_effective_length(const afx_oracle*, const double* x, int64_t n_samples, double* out3);

/* TStatistics restatements exposed for the reference's own known-answer tests */
double afx_oracle_sum(const double* x, int n);
double afx_oracle_mean(const double* x, int n);
double afx_oracle_variance(const double* x, int n, double mean);
double afx_oracle_geometric_mean(const double* x, int n);
double afx_oracle_centroid(const double* x, int n);
double afx_oracle_spread(const double* x, int n, double centroid);
double afx_oracle_skewness(const double* x, int n, double centroid, double spread);
double afx_oracle_kurtosis(const double* x, int n, double centroid, double spread);
double afx_oracle_flatness(const double* x, int n);
double afx_oracle_flatness_db(const double* x, int n);
double afx_oracle_correlation(const double* a, const double* b, int n);
double afx_oracle_median(const double* x, int n);
double afx_oracle_min(const double* x, int n);
double afx_oracle_max(const double* x, int n);
double afx_oracle_lin_to_db(double v);
/* TStatistics::Calc: out[13] = min,max,median,mean,gmean,variance,centroid,spread,skewness,
 * kurtosis,flatness,dmean,dvariance (fields the reference leaves untouched stay as passed in) */
void afx_oracle_calc_statistics(const double* x, int n, double* out13);

/* ---- LoadSample front end (SampleAnalyser.cpp:484-718) on already-decoded interleaved PCM ----
 * PINNED against oracle/_ref/ref_driver `load` (tests/golden/load.npz, bit-exact): SampleAnalyser.cpp itself
 * does not build here (Shark, LightGBM, CoreTypes), so the driver restates the member's flow around the
 * reference's own TSampleConverter conversions, TMathT<float>::GetMinMax, TAudioMath::DbToLin and constants.
 * format: 0 = int16, 1 = packed little-endian int24, 2 = float32 (the reference's decoders turn all
 * of them into "16-bit floats", CoreFileFormats/Export/SampleConverter.h:446-449, 474-486, 529-533). */
typedef struct {
  float peak_value;        /* TSampleData::mPeakValue */
  float rms_value;         /* TSampleData::mRmsValue  */
  int32_t data_offset;     /* TSampleData::mDataOffset = -leading + start pad */
  int32_t silent_leading;
  int32_t silent_trailing;
  int64_t n_samples;       /* size of TSampleData::mData */
} afx_oracle_load_info;
/* returns a malloc'ed normalised mono buffer (caller frees with afx_oracle_free) */
double* afx_oracle_load_sample(const void* pcm, int format, int channels, int64_t n_frames, int fft_size,
                               afx_oracle_load_info* info);

/* ---- the sample-rate conversion of LoadSample (SA:563-607): libresample 0.1.3 as the reference calls it ----
 * (afx_oracle_resample.c; PINNED against oracle/_ref/ref_driver `resample`, tests/golden/resample.npz, bit-exact)
 * in: the mono "16-bit float" buffer of a file at file_rate; returns malloc'ed out[*n_out], *n_out = NewSizeInSamples,
 * *n_written = the samples the converter produced (<= *n_out; the rest is 0). */
float* afx_oracle_resample(const float* in, int64_t n, int file_rate, int analyser_rate, int64_t* n_out, int64_t* n_written);
/* the filter's right wing, float[69632] (resample.c:110-124) */
void afx_oracle_resample_filter(float* imp);
/* afx_oracle_load_sample for a file at file_rate analysed at analyser_rate (the conversion runs on the mono mix,
 * before rms / peak / trim, as in SA:558-610) */
double* afx_oracle_load_sample_at(const void* pcm, int format, int channels, int64_t n_frames, int file_rate,
                                  int analyser_rate, int fft_size, afx_oracle_load_info* info);
void afx_oracle_free(void* p);

/* ---- rhythm tracker: the 512/128 loop behind the per-frame loop (SampleAnalyser.cpp:983-1048), afx_oracle_rhythm.c ----
 * PARTLY PINNED, see the header of afx_oracle_rhythm.c: the onset STFT front end and the aubio beat-tracking pass are
 * pinned against the reference's own objects (ref_driver `onsetfft`, `beattrack`); TOnsetDetector / TRhythmTracker /
 * TCannyWindow do not link here, their restatement is "parity unpinned". */
int afx_oracle_sample_rate(const afx_oracle*);
int64_t afx_oracle_analysed_length(const afx_oracle*, int64_t n_samples, int apply_cap);
int64_t afx_oracle_rhythm_frames(const afx_oracle*, int64_t n_samples, int apply_cap);
/* onsets [2][T] (0 = complex, 1 = percussive; TRhythmTracker::Onsets), optional sharpened [2][T] (Canny window) and
 * odf [2][T] (onset function before median removal); out14 = per type {onset_count, tempo, tempo_confidence,
 * onset_frequency_mean, onset_strength, onset_contrast}, then final_tempo, final_tempo_confidence.
 * original_rate / original_samples / data_offset: TSampleData::mOriginalSampleRate, mOriginalNumberOfSamples,
 * mDataOffset (SampleAnalyser.cpp:1001-1004).  Returns T. */
int64_t afx_oracle_run_rhythm(const afx_oracle*, const double* x, int64_t n_samples, int apply_cap,
                              int original_rate, int64_t original_samples, int data_offset, double* onsets,
                              double* sharpened, double* odf, double* out14);
/* TOnsetFftProcessor::LoadFrame on one 512-sample frame: DC, "Nyquist", magnitude[255], phase[255] as floats */
void afx_oracle_onset_polar(const double* x512, float* dc, float* nyquist, float* mag255, float* phase255);
/* one aubio_beattracking_do on a fresh tracker: bpm and the [taktik] confidence (beattracking.c) */
void afx_oracle_beattrack(const double* df, int winlen, int hop, int rate, double* bpm, double* confidence);
/* TCannyWindow(12, 16).Apply in place (CannyWindow.cpp:27-68) */
void afx_oracle_canny(double* x, int n);

#ifdef __cplusplus
}
#endif
#endif
