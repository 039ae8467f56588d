/* oracle/afx_oracle_rhythm.c -- TEST INFRASTRUCTURE ONLY (see afx_oracle.h).
 *
 * CPU restatement of the reference's rhythm tracker: the second, 512/128 loop of
 * TSampleAnalyser::AnalyzeLowLevelDescriptors (R/Source/Crawler/FeatureExtraction/Source/SampleAnalyser.cpp:983-1048).
 *   TRhythmTracker            R/Source/Crawler/FeatureExtraction/Source/RhythmTracker.cpp   ("RT.cpp")
 *   TOnsetFftProcessor/TOnsetDetector  R/Source/Core/AudioTypes/Source/OnsetDetector.cpp     ("OD.cpp")
 *   TCannyWindow              R/Source/Crawler/FeatureExtraction/Source/CannyWindow.cpp      ("CW.cpp")
 *   aubio beattracking        R/3rdParty/Aubio/Dist/src/tempo/beattracking.c                 ("bt.c"), mathutils.c
 *
 * Parity status: PARTLY PINNED.
 *   pinned   - onset STFT front end (window, FFT, magnitude, phase: afx_oracle_onset_polar) against the reference's own
 *              TFftWindow-formula / ooura_cdft / TAudioMath::Magnitude / TAudioMath::Phase objects (ref_driver `onsetfft`);
 *            - tempo and confidence of one beat-tracking pass (afx_oracle_beattrack) against the reference's own aubio
 *              beattracking.c object (ref_driver `beattrack`);
 *            - TStatistics::Mean / Variance (already pinned, afx_oracle.c).
 *   UNPINNED - the whitening follower, the two onset functions, median removal / detection, Canny sharpening, peak,
 *              strength, contrast and the duration heuristics: TOnsetDetector / TRhythmTracker / TCannyWindow do not
 *              link here (TArray/TList need TMemory -> TSystem::InMain(): LinuxSystem.cpp includes the <sys/sysctl.h> this
 *              glibc no longer ships; TString needs libiconv, in the reference tree only as a git-LFS pointer: oracle/Makefile),
 *              so these are restated from the sources cited below and checked only by hand-computed cases in tests/.
 *
 * Float semantics: the reference keeps the polar spectrum, the whitening output and the onset functions in `float`
 * (OD.cpp); every expression below uses the same type as its reference counterpart (build with -ffp-contract=off).
 */
#include "afx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RT_FFT 512          /* TempoFftSize, SA.cpp:985 */
#define RT_HOP 128          /* TempoHopSize, SA.cpp:986 */
#define RT_BINS 255         /* mNumbins = fft/2 - 1, OD.cpp:43 */

static const double kPi = 3.1415926535897932384626433832795;      /* MPi,   InlineMath.h:13 */
static const double k2Pi = 6.2831853071795864769252867665590;     /* M2Pi,  InlineMath.h:16 */
static const double kInv2Pi = 0.15915494309189533576888376337251; /* MInv2Pi, InlineMath.h:17 */

/* ---- onset STFT front end (OD.cpp:116-160) ------------------------------------------------- */

static void fft512(double* re, double* im) { /* X[k] = sum x[j] e^{+2 pi i jk/N}, no scaling (kNoDiv, OD.cpp:56) */
  static double twr[RT_FFT / 2], twi[RT_FFT / 2];
  static int ready = 0;
  const int n = RT_FFT;
  int i, j, len;
  if (!ready) {
    for (i = 0; i < n / 2; ++i) { twr[i] = cos(k2Pi * i / n); twi[i] = sin(k2Pi * i / n); }
    ready = 1;
  }
  for (i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = n / len;
    for (i = 0; i < n; i += len)
      for (j = 0; j < half; ++j) {
        const double wr = twr[j * step], wi = twi[j * step];
        const double ur = re[i + j], ui = im[i + j], xr = re[i + j + half], xi = im[i + j + half];
        const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
        re[i + j] = ur + vr; im[i + j] = ui + vi;
        re[i + j + half] = ur - vr; im[i + j + half] = ui - vi;
      }
  }
}

/* TOnsetFftProcessor::LoadFrame: Hanning window w[i] = 0.5 (1 - cos(2 pi i / (N-1))) (Fourier.cpp:505, 545-551),
 * complex FFT of the real frame, then "bins" 0..254 of the COMPLEX spectrum: mBin[i] is bin i (bin 0 = DC is included),
 * mDC = Re[0] and mNyquist = Im[0] -- the imaginary part of bin 0, not bin 256 (OD.cpp:147-151). */
void afx_oracle_onset_polar(const double* x, float* dc, float* nyquist, float* mag, float* phase) {
  double re[RT_FFT], im[RT_FFT];
  const double delta = 1.0 / (double)(RT_FFT - 1);
  int i;
  for (i = 0; i < RT_FFT; ++i) {
    re[i] = (0.5 * (1.0 - cos(k2Pi * (double)i * delta))) * x[i];
    im[i] = 0.0;
  }
  fft512(re, im);
  *dc = (float)re[0];
  *nyquist = (float)im[0];
  for (i = 0; i < RT_BINS; ++i) {
    mag[i] = (float)sqrt(re[i] * re[i] + im[i] * im[i]);   /* AudioMath.cpp:497-503 */
    phase[i] = (float)atan2(im[i], re[i]);                 /* AudioMath.cpp:637-643 */
  }
}

/* ---- detectors ------------------------------------------------------------------------------ */

static float phase_rewrap(float p) { /* SPhaseRewrap, OD.cpp:18-22 */
  return (p > -(float)kPi && p < (float)kPi) ? p
                                              : p + (float)k2Pi * (1.f + floorf((-(float)kPi - p) * (float)kInv2Pi));
}

typedef struct {
  int power;              /* 1: kFunctionPower, 0: kFunctionRComplex */
  int medspan, mingap, gapleft;
  float thresh, normfactor, odfparam;
  float* odfvals;         /* [medspan], newest first */
  float other[3 * RT_BINS];
  float post, postprev;
} rt_detector;

static int cmp_float(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

static void detector_init(rt_detector* d, int power, float rate, float thresh, float medspan_s, float mingap_s) {
  memset(d, 0, sizeof(*d));
  d->power = power;
  d->thresh = thresh;
  d->medspan = (int)((rate * medspan_s) / (float)RT_HOP + 0.5f);   /* OD.cpp:262-264 */
  if (d->medspan < 3) d->medspan = 3;
  d->mingap = (int)((rate * mingap_s) / (float)RT_HOP + 0.5f);     /* OD.cpp:272 */
  d->odfvals = (float*)calloc((size_t)d->medspan, sizeof(float));
  d->odfparam = 0.01f;
  d->normfactor = power ? 2560.f / (float)((RT_BINS + 2) * RT_FFT)               /* OD.cpp:277-279 */
                        : (float)(231.70475 / pow((double)RT_FFT, 1.5));         /* OD.cpp:300-302 */
}

/* TOnsetDetector::Process = CalculateOnsetFunction + DetectOnset (OD.cpp:356-587); returns the detection flag */
static int detector_process(rt_detector* d, float dc, float nyquist, const float* mag, const float* phase, float* raw_odf) {
  float* v = d->odfvals;
  float* sorted;
  float median;
  int i, detected;
  memmove(v + 1, v, (size_t)(d->medspan - 1) * sizeof(float));   /* OD.cpp:375 */
  if (d->power) {                                                 /* OD.cpp:380-388 */
    *v = (nyquist * nyquist) + (dc * dc);
    for (i = 0; i < RT_BINS; ++i) {
      const float m = mag[i];
      *v += m * m;
    }
  } else {                                                        /* OD.cpp:398-458, Rectify = true */
    double total = 0.0;
    for (i = 0; i < RT_BINS; ++i) {
      const float cur = fabsf(mag[i]);
      const float pred_mag = d->other[3 * i], yester_phase = d->other[3 * i + 1], yester_diff = d->other[3 * i + 2];
      if (cur > d->odfparam) {
        if (!(cur < pred_mag)) {
          const float pred_phase = yester_phase + yester_diff;
          float dev = pred_phase - phase[i];
          dev = sqrtf(pred_mag * pred_mag + cur * cur - pred_mag * cur * cosf(phase_rewrap(dev)));
          total += dev;
        }
      }
    }
    for (i = 0; i < RT_BINS; ++i) {
      float diff;
      d->other[3 * i] = fabsf(mag[i]);
      diff = phase[i] - d->other[3 * i + 1];
      d->other[3 * i + 1] = phase[i];
      d->other[3 * i + 2] = phase_rewrap(diff);
    }
    *v = (float)total;
  }
  v[0] *= d->normfactor;                                          /* OD.cpp:544 */
  if (raw_odf) *raw_odf = v[0];
  /* DetectOnset, OD.cpp:549-587 */
  d->postprev = d->post;
  sorted = (float*)malloc((size_t)d->medspan * sizeof(float));
  memcpy(sorted, v, (size_t)d->medspan * sizeof(float));
  qsort(sorted, (size_t)d->medspan, sizeof(float), cmp_float);
  median = (d->medspan & 1) ? sorted[(d->medspan - 1) >> 1]
                            : ((sorted[d->medspan >> 1] + sorted[(d->medspan >> 1) - 1]) * 0.5f);
  free(sorted);
  d->post = v[0] - median;
  if (d->gapleft != 0) {
    d->gapleft--;
    detected = 0;
  } else {
    detected = (d->post > d->thresh) && (d->postprev <= d->thresh);
    if (detected) d->gapleft = d->mingap;
  }
  return detected;
}

/* ---- sharpening, peaks, scalars ------------------------------------------------------------- */

#define RT_CANNY 12          /* MCannyWindowLength, RT.cpp:37 */
#define RT_CANNY_SHAPE 16.0  /* MCannyWindowShape,  RT.cpp:38 */
#define RT_PEAK_WINDOW 24    /* MPeakWindowLength,  RT.cpp:41 */
#define RT_PEAK_THRESHOLD 0.1

void afx_oracle_canny(double* x, int n) { /* TCannyWindow::Apply, CW.cpp:27-68; window CW.cpp:72-78 */
  double win[2 * RT_CANNY + 1];
  double* tmp = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  const double sq = RT_CANNY_SHAPE * RT_CANNY_SHAPE;
  double mean, var;
  int i, s;
  for (i = -RT_CANNY; i < RT_CANNY + 1; ++i) win[i + RT_CANNY] = (double)i / sq * exp(-1.0 * (i * i) / (2.0 * sq));
  for (i = 0; i < n; ++i) {
    double sum = 0.0;
    for (s = -RT_CANNY; s < RT_CANNY; ++s)
      if (i + s >= 0 && i + s < n) sum += x[i + s] * win[s + RT_CANNY];
    tmp[i] = sum;
  }
  memcpy(x, tmp, sizeof(double) * (size_t)n);
  free(tmp);
  mean = afx_oracle_mean(x, n);
  var = afx_oracle_variance(x, n, mean);
  if (var > 0.0) {
    const double sd = sqrt(var);
    for (i = 0; i < n; ++i) { const double z = (x[i] - mean) / sd; x[i] = 0.0 > z ? 0.0 : z; }   /* MMax(0.0, z) */
  }
}

static int is_window_peak(const double* x, int n, int i) { /* RT.cpp:362-372, 643-653 */
  int j;
  for (j = -RT_PEAK_WINDOW; j <= RT_PEAK_WINDOW; ++j)
    if (i + j >= 0 && i + j < n && x[i + j] > x[i]) return 0;
  return 1;
}

static int calculate_peaks(const double* x, int n, double* peaks) { /* RT.cpp:624-660 */
  int i, c = 0;
  for (i = 0; i < n; ++i) {
    if (x[i] <= RT_PEAK_THRESHOLD) continue;
    if (is_window_peak(x, n, i)) peaks[c++] = x[i];
  }
  return c;
}

static int cmp_dbl(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

static double rhythm_contrast(const double* x, int n) { /* RT.cpp:325-412 */
  double* sorted = (double*)malloc(sizeof(double) * (size_t)n * 3);
  double* peaks = sorted + n;
  double* valleys = peaks + n;
  double threshold, valley_value, total_mean, peak_mean, valley_mean, r = 0.0;
  int i, np = 0, valley_pos = 0;
  memcpy(sorted, x, sizeof(double) * (size_t)n);
  qsort(sorted, (size_t)n, sizeof(double), cmp_dbl);
  threshold = sorted[(int)(85.0 / 100.0 * (n - 1))];
  valley_value = threshold;
  for (i = 0; i < n; ++i) {
    if (x[i] < valley_value) { valley_pos = i; valley_value = x[i]; }
    if (x[i] < threshold) continue;
    if (is_window_peak(x, n, i)) {
      peaks[np] = x[i];
      valleys[np] = x[valley_pos];
      ++np;
      valley_value = x[i];
    }
  }
  total_mean = afx_oracle_mean(x, n);
  peak_mean = afx_oracle_mean(peaks, np);
  valley_mean = afx_oracle_mean(valleys, np) + 0.0001;
  if (peak_mean != 0.0) r = -1.0 * pow(peak_mean / valley_mean, 1.0 / log(total_mean + 0.0001));
  free(sorted);
  return r;
}

/* ---- aubio beat tracking: one aubio_beattracking_do on a fresh tracker (bt.c:59-118, 132-186, 273-404, 424-444) ---- */

static double quadratic_peak_pos(const double* x, unsigned n, unsigned pos) { /* mathutils.c:494-506 */
  double s0, s1, s2;
  if (pos == 0 || pos == n - 1) return pos;
  s0 = x[pos - 1]; s1 = x[pos]; s2 = x[pos + 1];
  return pos + .5 * (s0 - s2) / (s0 - 2. * s1 + s2);
}

static double quadratic_peak_mag(const double* x, unsigned n, double pos) { /* mathutils.c:508-517 */
  const unsigned index = (unsigned)(pos - .5) + 1;
  if (pos >= n || pos < 0.) return 0.;
  if ((double)index == pos) return x[index];
  return x[index] - .25 * (x[index - 1] - x[index + 1]) * (pos - index);
}

void afx_oracle_beattrack(const double* df, int winlen_in, int hop, int rate, double* bpm, double* confidence) {
  const unsigned winlen = (unsigned)winlen_in, laglen = winlen / 4;
  const double rayparam = 60. * rate / 120. / hop;          /* bt.c:65 */
  const unsigned rayparam_u = (unsigned)rayparam;           /* p->rayparam is uint_t, bt.c:46, 82 */
  double* acf = (double*)calloc((size_t)winlen + laglen + 1, sizeof(double));
  double* acfout = acf + winlen;
  double rp, bp, acf_sum = 0.0;
  unsigned i, j, a, b, maxindex = 0;
  double tmp = 0.0;
  for (i = 0; i < winlen; ++i) {                            /* aubio_autocorr, mathutils.c:652-666 */
    double t = 0.;
    for (j = i; j < winlen; ++j) t += df[j - i] * df[j];
    acf[i] = t / (double)(winlen - i);
  }
  if (laglen >= 2)
    for (i = 1; i < laglen - 1; ++i)                        /* shift invariant comb filterbank, bt.c:165-172 */
      for (a = 1; a <= 4; ++a)
        for (b = 1; b < 2 * a; ++b) acfout[i] += acf[i * a + b - 1] * 1. / (2. * a - 1.);
  for (i = 0; i < laglen; ++i)                              /* Rayleigh weight, bt.c:105-108, 174 */
    acfout[i] *= ((double)(i + 1.) / (rayparam * rayparam)) * exp((-((i + 1.) * (i + 1.)) / (2. * (rayparam * rayparam))));
  for (j = 0; j < laglen; ++j) {                            /* fvec_max_elem, mathutils.c:268-283 */
    maxindex = (tmp > acfout[j]) ? maxindex : j;
    tmp = (tmp > acfout[j]) ? tmp : acfout[j];
  }
  if (maxindex > 0 && laglen > 0 && maxindex < laglen - 1) rp = quadratic_peak_pos(acfout, laglen, maxindex);
  else rp = rayparam_u;                                     /* bt.c:177-182 */
  /* aubio_beattracking_checkstate on the zero-initialised state: gp = 0, timesig = 0 -> bp = rp (bt.c:362-367),
   * then doubled while 0 < bp < 25 (bt.c:372-379) */
  bp = rp;
  while (0 < bp && bp < 25) bp = bp * 2;
  *bpm = (bp != 0) ? 60. / (hop * bp / (double)rate) : 0.;  /* bt.c:412-432 */
  for (i = 0; i < laglen; ++i) acf_sum += acfout[i];        /* bt.c:435-444 ([taktik] confidence) */
  /* the reference function has no return for acf_sum == 0 (undefined value); 0 is used here */
  *confidence = (acf_sum != 0.) ? quadratic_peak_mag(acfout, laglen, rp) / acf_sum : 0.;
  free(acf);
}

/* ---- TRhythmTracker ------------------------------------------------------------------------- */

static const double kOnsetThreshold[2] = {0.2, 0.8};   /* MComplexOnsetThreshold, MPercussiveOnsetThreshold, RT.cpp:26, 32 */

static int onset_count(const double* onsets, int n, int type) { /* RT.cpp:124-137 */
  int i, c = 0;
  for (i = 0; i < n; ++i) if (onsets[i] > kOnsetThreshold[type]) ++c;
  return c;
}

static double calculate_tempo(const double* raw, const double* sharp, int n, int rate, int type, double* confidence) {
  double tempo, conf;                                     /* RT.cpp:159-234 */
  if (onset_count(raw, n, type) < 4) { *confidence = 0.0; return 0.0; }
  afx_oracle_beattrack(sharp, n, RT_HOP, rate, &tempo, &conf);
  conf = conf * 16.0;
  conf = conf < 0.0 ? 0.0 : (conf > 1.0 ? 1.0 : conf);   /* MClip */
  if (tempo < 20.0 || tempo > 300.0) { *confidence = 0.0; return 0.0; }
  while (tempo < 80.0) tempo *= 2.0;
  while (tempo >= 200.0) tempo /= 2.0;
  *confidence = conf;
  return tempo;
}

static double guess_number_of_beats(double min_bpm, double duration) { /* RT.cpp:434-486 */
  const double beats_per_bar = 4.0, max_beat = 60.0 / min_bpm, max_bar = beats_per_bar * max_beat;
  int divider, bars;
  if (duration < max_beat) return 0.0f;
  if (duration < max_bar) {
    for (divider = 4 / 2; divider >= 1; divider /= 2) {
      const double divided = ((double)4 / (double)divider) * max_beat;
      if (4 % divider == 0 && duration < divided) return (float)(beats_per_bar / (double)divider);
    }
    return (float)beats_per_bar;
  }
  for (bars = 1; bars <= 8; bars *= 2) {
    const double beats = beats_per_bar * bars;
    if (duration / beats < max_beat) return (float)beats;
  }
  return 0.0f;
}

static int ms_to_samples_f(int rate, float ms) { /* TAudioMath::MsToSamples, AudioMath.inl:127-130 */
  const float v = (float)rate / 1000.0f * ms;   /* f2iRound: truncate v + sign(v) / 2, InlineMath.inl:758-761 */
  return (int)(v + (signbit(v) ? -0.5f : 0.5f));
}

static double onset_match_confidence(const double* raw, int n, int rate, double offset_s, double beats, double tempo,
                                     int type) { /* RT.cpp:490-540 */
  const int offset_samples = ms_to_samples_f(rate, (float)(offset_s * 1000));
  const double samples_per_beat = 60.0 / tempo * rate;
  const int range = (int)(samples_per_beat / 32) / RT_HOP;
  double strength = 0;
  int i, j;
  for (i = 0; i < beats * 2; ++i) {
    const int t = (int)(i * samples_per_beat / 2.0) + offset_samples;
    const int idx = (t + RT_HOP / 2) / RT_HOP;
    double peak = 0.0;
    for (j = idx - range; j < idx + range; ++j)
      if (j >= 0 && j < n) peak = peak > raw[j] ? peak : raw[j];
    if (peak >= kOnsetThreshold[type]) strength += 1.0;
  }
  {
    const double r = strength / (beats * 2) * 2.0;
    return 1.0 < r ? 1.0 : r;
  }
}

static double tempo_with_heuristics(const double* raw, const double* sharp, int n, int rate, int type, double tempo_in,
                                    double conf_in, double duration_s, double offset_s, double* confidence) {
  double tempo = tempo_in, samples_per_beat;               /* RT.cpp:238-325 */
  int last;
  if (tempo_in == 0) { *confidence = 0.0; return 0.0; }
  *confidence = conf_in;
  samples_per_beat = 60.0 / tempo * rate;
  last = n - 1;
  while (last > 0 && sharp[last] < RT_PEAK_THRESHOLD) --last;
  if ((double)(last * RT_HOP) < samples_per_beat * 3) { *confidence = 0.0; return 0.0; }
  {
    const double beats = guess_number_of_beats(80, duration_s);   /* MaxBpm is unused by the reference function */
    if (beats >= 4 && beats <= 16) {
      const double guessed_bpm = beats / (duration_s / 60);
      const double delay = (double)((float)(RT_HOP / 2) / ((float)rate / 1000.0f)) / 1000.0;  /* SamplesToMs, AudioMath.inl:134-137 */
      const double g = onset_match_confidence(raw, n, rate, offset_s + delay, beats, guessed_bpm, type);
      if ((g > 0.5) || (*confidence < 0.1 && g > 0.1) || (*confidence < 0.5 && fabs(guessed_bpm - tempo) < 10)) {
        tempo = guessed_bpm;
        *confidence = 0.5 > g ? 0.5 : g;
      }
    }
  }
  return tempo;
}

int64_t afx_oracle_rhythm_frames(const afx_oracle* o, int64_t n_samples, int apply_cap) { /* SA.cpp:760-764, 991 */
  const int64_t len = afx_oracle_analysed_length(o, n_samples, apply_cap);
  int64_t f = 0, n;
  for (n = 0; (n + RT_FFT - 1) < len; n += RT_HOP) ++f;
  return f;
}

int64_t afx_oracle_run_rhythm(const afx_oracle* o, const double* x, int64_t n_samples, int apply_cap,
                              int original_rate, int64_t original_samples, int data_offset, double* onsets,
                              double* sharpened, double* odf, double* out14) {
  const int rate = afx_oracle_sample_rate(o);
  /* SA.cpp:1001-1004: TAudioMath::SamplesToMs (float, AudioMath.inl:134-137) divided by the int 1000 -> float */
  const double duration_s = (double)(((float)(int)original_samples / ((float)original_rate / 1000.0f)) / 1000);
  const double offset_s = (double)(((float)data_offset / ((float)original_rate / 1000.0f)) / 1000);
  const int64_t T = afx_oracle_rhythm_frames(o, n_samples, apply_cap);
  const int n = (int)T;
  /* TOnsetFftProcessor state (OD.cpp:25-83): whitening follower psp[257], relax coefficient in float */
  double psp[RT_BINS + 2];
  const float relax_time = (float)25.0, floor_ = 0.1f;                                        /* RT.cpp:21, OD.cpp:47 */
  const float coef = (float)(exp((-2.30258509 * (float)RT_HOP) / (relax_time * (float)rate))); /* OD.cpp:110-112 */
  rt_detector det[2];
  double* sharp[2];
  double* peaks;
  double conf[2], tempo[2];
  int64_t f;
  int t, i;
  memset(out14, 0, sizeof(double) * 14);
  if (n <= 0) return 0;
  memset(psp, 0, sizeof(psp));
  detector_init(&det[0], 0, (float)rate, (float)0.2, (float)0.2, (float)0.06);   /* RT.cpp:24-28, 68-78 */
  detector_init(&det[1], 1, (float)rate, (float)0.8, (float)0.2, (float)0.12);   /* RT.cpp:30-34, 80-90 */
  for (f = 0; f < T; ++f) {
    float dc, ny, mag[RT_BINS], phase[RT_BINS];
    double value, old;
    afx_oracle_onset_polar(x + f * RT_HOP, &dc, &ny, mag, phase);
    /* Whiten, kWhiteningAdaptMax (OD.cpp:186-240): psp[0] follows |DC|, psp[256] |Nyquist|, psp[1 + i] bin i */
    value = fabsf(dc); old = psp[0];
    if (value < old) value = value + (old - value) * coef;
    psp[0] = value;
    value = fabsf(ny); old = psp[1 + RT_BINS];
    if (value < old) value = value + (old - value) * coef;
    psp[1 + RT_BINS] = value;
    for (i = 0; i < RT_BINS; ++i) {
      value = fabsf(mag[i]); old = psp[1 + i];
      if (value < old) value = value + (old - value) * coef;
      psp[1 + i] = value;
    }
    dc /= (float)((double)floor_ > psp[0] ? (double)floor_ : psp[0]);
    ny /= (float)((double)floor_ > psp[1 + RT_BINS] ? (double)floor_ : psp[1 + RT_BINS]);
    for (i = 0; i < RT_BINS; ++i) mag[i] /= (float)((double)floor_ > psp[1 + i] ? (double)floor_ : psp[1 + i]);
    for (t = 0; t < 2; ++t) {                                                    /* RT.cpp:107-120 */
      float raw;
      const int detected = detector_process(&det[t], dc, ny, mag, phase, &raw);
      onsets[(size_t)t * n + f] = detected ? (double)det[t].post : 0.0;
      if (odf) odf[(size_t)t * n + f] = (double)raw;
    }
  }
  free(det[0].odfvals);
  free(det[1].odfvals);
  /* scalars, SA.cpp:1006-1048 */
  sharp[0] = (double*)malloc(sizeof(double) * (size_t)n * 3);
  sharp[1] = sharp[0] + n;
  peaks = sharp[1] + n;
  for (t = 0; t < 2; ++t) {
    const double* raw = onsets + (size_t)t * n;
    double* o6 = out14 + 6 * t;
    int np;
    memcpy(sharp[t], raw, sizeof(double) * (size_t)n);
    afx_oracle_canny(sharp[t], n);                                               /* RT.cpp:603-617 */
    if (sharpened) memcpy(sharpened + (size_t)t * n, sharp[t], sizeof(double) * (size_t)n);
    o6[0] = onset_count(raw, n, t);
    tempo[t] = o6[1] = calculate_tempo(raw, sharp[t], n, rate, t, &conf[t]);
    o6[2] = conf[t];
    np = calculate_peaks(sharp[t], n, peaks);
    o6[3] = (double)np / (double)n * (double)RT_HOP / (double)RT_FFT;            /* RT.cpp:288-289 */
    if (np) {                                                                    /* RT.cpp:316-318 */
      const double m = afx_oracle_mean(peaks, np) / 4.0;
      o6[4] = m < 0.0 ? 0.0 : (m > 1.0 ? 1.0 : m);
    }
    o6[5] = rhythm_contrast(sharp[t], n);
  }
  t = (conf[1] > conf[0]) ? 1 : 0;                                               /* SA.cpp:1029-1048 */
  out14[12] = tempo_with_heuristics(onsets + (size_t)t * n, sharp[t], n, rate, t, tempo[t], conf[t], duration_s,
                                    offset_s, &out14[13]);
  free(sharp[0]);
  return T;
}
