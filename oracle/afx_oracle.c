/* oracle/afx_oracle.c -- TEST INFRASTRUCTURE ONLY (see afx_oracle.h).
 *
 * From-scratch plain-C restatement of the AFEC low-level per-frame pipeline.  Every function
 * cites the reference file:line it follows ("R/" = /root/reference/).  All arithmetic is IEEE
 * double, scalar, in the reference's summation order, compiled with -ffp-contract=off.
 *
 * Abbreviations: SA.cpp = R/Source/Crawler/FeatureExtraction/Source/SampleAnalyser.cpp,
 * Stat.cpp = R/Source/Crawler/FeatureExtraction/Source/Statistics.cpp,
 * Xt/ = R/3rdParty/LibXtract/Dist/.
 */
#include "afx_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define N_CEP 14   /* kNumberOfCepstrumCoefficients, SampleDescriptors.h:464 */
#define N_BANDS 28 /* kNumberOfSpectrumBands */
#define N_SUB 14   /* kNumberOfSpectrumSubBands */

/* R/Source/Core/CoreTypes/Export/InlineMath.h:32 -- a *float* literal widened to double */
static const double kEpsilon = (double)1e-12f;

struct afx_oracle {
  int sample_rate, fft, hop;
  int first_bin, last_bin, bin_count; /* SA.cpp:171-175 */
  double* window;                     /* [fft]          */
  double* mel;                        /* [N_CEP][fft/2] */
  double* tw_re; double* tw_im;       /* FFT twiddles   */
};

/* TMath::d2iRound, R/Source/Core/CoreTypes/Export/InlineMath.inl:823-826 (truncate x + sign/2) */
static int d2i_round(double v) { return (int)(v + ((v < 0.0) ? -0.5 : 0.5)); }

/* TAudioMath::MsToSamples, R/Source/Core/AudioTypes/Export/AudioMath.inl:125-128 (float maths) */
static int ms_to_samples(int rate, float ms) {
  float v = (float)rate / 1000.0f * ms;
  return (int)(v + ((v < 0.0f) ? -0.5f : 0.5f));
}

/* ---- tables ------------------------------------------------------------------------------ */

/* xtract_init_window(N, XTRACT_HANN) -> hann(), Xt/src/window.c:67-76; then x2 (SA.cpp:178-181) */
static void make_window(double* w, int n) {
  const double M = n - 1;
  for (int i = 0; i < n; ++i) w[i] = 0.5 * (1.0 - cos(2.0 * M_PI * (double)i / M));
  for (int i = 0; i < n; ++i) w[i] *= 2.0;
}

/* xtract_init_mfcc, Xt/src/init.c:237-382, XTRACT_EQUAL_GAIN branch; called as
 * (N = fft/2, nyquist = sample_rate/2, 20, 15500, 14) at SA.cpp:195-197.
 * Quirks kept: M = N>>1; integer truncation of the peaks; division by fft_peak[0]==0 -> inf
 * (only table[0][0]=0 is written with it); the running index i carried across filters. */
static void make_mel(double* tab, int N, double nyquist, double fmin, double fmax, int nb) {
  double mel_max = 1127 * log(1 + fmax / 700);
  double mel_min = 1127 * log(1 + fmin / 700);
  double bw = (mel_max - mel_min) / nb;
  double* mel_peak = (double*)malloc((nb + 2) * sizeof(double));
  double* lin_peak = (double*)malloc((nb + 2) * sizeof(double));
  int* fft_peak = (int*)malloc((nb + 2) * sizeof(int));
  int M = N >> 1, n, i, k;
  mel_peak[0] = mel_min;
  lin_peak[0] = fmin;
  fft_peak[0] = (int)(lin_peak[0] / nyquist * M);
  for (n = 1; n < nb + 2; ++n) {
    mel_peak[n] = mel_peak[n - 1] + bw;
    lin_peak[n] = 700 * (exp(mel_peak[n] / 1127) - 1);
    fft_peak[n] = (int)(lin_peak[n] / nyquist * M);
  }
  i = 0;
  for (n = 0; n < nb; ++n) {
    double* t = tab + (size_t)n * N;
    const double height = 1.0; /* EQUAL_GAIN: height * norm_fact = 1 */
    double inc, val;
    if (n == 0) inc = height / fft_peak[n];
    else inc = height / (fft_peak[n] - fft_peak[n - 1]);
    val = 0;
    for (k = 0; k < i; ++k) t[k] = 0.0;
    for (; i <= fft_peak[n]; ++i) { t[i] = val; val += inc; }
    inc = height / (fft_peak[n + 1] - fft_peak[n]);
    val = 0;
    {
      int next_peak = fft_peak[n + 1];
      for (i = next_peak; i > fft_peak[n]; --i) { t[i] = val; val += inc; }
      for (k = next_peak + 1; k < N; ++k) t[k] = 0.0;
    }
  }
  free(mel_peak); free(lin_peak); free(fft_peak);
}

afx_oracle* afx_oracle_create(int sample_rate, int fft_size, int hop_size) {
  afx_oracle* o;
  if (fft_size < 8 || (fft_size & (fft_size - 1)) || hop_size <= 0) return NULL;
  o = (afx_oracle*)calloc(1, sizeof(*o));
  o->sample_rate = sample_rate; o->fft = fft_size; o->hop = hop_size;
  {
    /* SA.cpp:171-175: int / int stored in a double */
    const double fpb = (double)(sample_rate / fft_size);
    o->first_bin = d2i_round(20.0 / fpb);
    o->last_bin = d2i_round(15500.0 / fpb);
    o->bin_count = o->last_bin - o->first_bin + 1;
  }
  o->window = (double*)malloc(sizeof(double) * fft_size);
  make_window(o->window, fft_size);
  o->mel = (double*)malloc(sizeof(double) * N_CEP * (fft_size / 2));
  make_mel(o->mel, fft_size / 2, (double)(sample_rate / 2), 20.0, 15500.0, N_CEP);
  o->tw_re = (double*)malloc(sizeof(double) * fft_size / 2);
  o->tw_im = (double*)malloc(sizeof(double) * fft_size / 2);
  for (int k = 0; k < fft_size / 2; ++k) {
    /* forward sign +, R/Source/Core/AudioTypes/Source/OouraFFT8g.cpp:36-39 */
    o->tw_re[k] = cos(2.0 * M_PI * k / fft_size);
    o->tw_im[k] = sin(2.0 * M_PI * k / fft_size);
  }
  return o;
}

void afx_oracle_destroy(afx_oracle* o) {
  if (!o) return;
  free(o->window); free(o->mel); free(o->tw_re); free(o->tw_im); free(o);
}
const double* afx_oracle_window(const afx_oracle* o) { return o->window; }
const double* afx_oracle_mel(const afx_oracle* o) { return o->mel; }
int afx_oracle_first_bin(const afx_oracle* o) { return o->first_bin; }
int afx_oracle_bin_count(const afx_oracle* o) { return o->bin_count; }

/* ---- STFT --------------------------------------------------------------------------------- */

/* The reference runs Ooura's split-radix cdft on the N-point complex signal (Re = windowed frame,
 * Im = 0), X[k] = sum x[j] e^{+2 pi i jk/N}, then scales by 1.0f/N
 * (R/Source/Core/AudioTypes/Source/Fourier.cpp:243-270; OouraFFT8g.cpp:36-39,289).  This is the
 * same DFT computed with an iterative radix-2 decimation-in-time FFT; bins agree with Ooura's to
 * a few ulp of the frame's largest bin. */
static void fft_forward(const afx_oracle* o, double* re, double* im) {
  const int n = o->fft;
  int i, j, len;
  for (i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = n / len;
    for (i = 0; i < n; i += len) {
      for (j = 0; j < half; ++j) {
        const double wr = o->tw_re[j * step], wi = o->tw_im[j * step];
        const double ur = re[i + j], ui = im[i + j];
        const double xr = re[i + j + half], xi = im[i + j + half];
        const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
        re[i + j] = ur + vr; im[i + j] = ui + vi;
        re[i + j + half] = ur - vr; im[i + j + half] = ui - vi;
      }
    }
  }
  {
    const double scale = 1.0f / n; /* Fourier.cpp:265-270 (kDivFwdByN) */
    for (i = 0; i < n; ++i) { re[i] *= scale; im[i] *= scale; }
  }
}

/* SA.cpp:826-845: xtract_windowed (Xt/src/helper.c:36-50) -> FFT -> TAudioMath::Magnitude
 * (R/Source/Core/AudioTypes/Source/AudioMath.cpp:497-503) on fft/2 bins; upper half cleared. */
static void stft_frame(const afx_oracle* o, const double* x, double* re, double* im, double* mag) {
  const int n = o->fft;
  int i;
  for (i = 0; i < n; ++i) { re[i] = x[i] * o->window[i]; im[i] = 0.0; }
  fft_forward(o, re, im);
  /* TAudioMath::Magnitude runs with the SSE DAZ + FZ bits set (AudioMath.cpp:25-35, 478-480): a product that
   * would be denormal is flushed to zero, so bins below ~1.5e-154 come out as exactly 0 */
  for (i = 0; i < n / 2; ++i) {
    double a = re[i] * re[i], b = im[i] * im[i], p;
    if (a < DBL_MIN) a = 0.0;
    if (b < DBL_MIN) b = 0.0;
    p = a + b;
    mag[i] = sqrt(p < DBL_MIN ? 0.0 : p);
  }
  for (i = n / 2; i < n; ++i) mag[i] = 0.0;
}

/* ---- TStatistics -------------------------------------------------------------------------- */

double afx_oracle_sum(const double* x, int n) { /* Stat.cpp:236-245 */
  double s = 0.0; for (int i = 0; i < n; ++i) s += x[i]; return s;
}
double afx_oracle_mean(const double* x, int n) { /* Stat.cpp:249-266 */
  if (n >= 2) return afx_oracle_sum(x, n) / (double)n;
  return (n == 1) ? x[0] : 0.0;
}
double afx_oracle_variance(const double* x, int n, double mean) { /* Stat.cpp:275-300 */
  if (n >= 2) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += (x[i] - mean) * (x[i] - mean);
    return r / n;
  }
  return 0.0;
}
double afx_oracle_min(const double* x, int n) { /* Stat.cpp:94-113 */
  if (n <= 0) return 0.0;
  double m = x[0]; for (int i = 1; i < n; ++i) m = (x[i] < m) ? x[i] : m; return m;
}
double afx_oracle_max(const double* x, int n) { /* Stat.cpp:117-136 */
  if (n <= 0) return 0.0;
  double m = x[0]; for (int i = 1; i < n; ++i) m = (x[i] > m) ? x[i] : m; return m;
}
static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
/* Stat.cpp:316-413: quickselect of element (0 + n-1)/2 == the lower median of the sorted array */
double afx_oracle_median(const double* x, int n) {
  if (n >= 2) {
    double* t = (double*)malloc(sizeof(double) * n);
    double m;
    memcpy(t, x, sizeof(double) * n);
    qsort(t, n, sizeof(double), cmp_double);
    m = t[(n - 1) / 2];
    free(t);
    return m;
  }
  return (n == 1) ? x[0] : 0.0;
}
double afx_oracle_geometric_mean(const double* x, int n) { /* Stat.cpp:417-455 */
  if (n >= 2) {
    const double too_large = 1.e64, too_small = 1.e-64;
    double sum_log = 0.0, product = 1.0;
    for (int i = 0; i < n; ++i) {
      product *= (fabs(x[i]) + 1e-20);
      if (product > too_large || product < too_small) { sum_log += log(product); product = 1.0; }
    }
    return exp((sum_log + log(product)) / (double)n);
  }
  return (n == 1) ? x[0] : 0.0;
}
double afx_oracle_centroid(const double* x, int n) { /* Stat.cpp:459-477 */
  const double s = afx_oracle_sum(x, n);
  if (s == 0.0) return 0.0;
  double sc = 0.0;
  for (int j = 0; j < n; ++j) sc += (double)j * x[j];
  return sc / s;
}
double afx_oracle_spread(const double* x, int n, double c) { /* Stat.cpp:486-506 */
  const double s = afx_oracle_sum(x, n);
  if (s == 0.0) return 0.0;
  double sc = 0.0;
  for (int j = 0; j < n; ++j) { const double t = j - c; sc += t * t * x[j]; }
  return sc / s;
}
double afx_oracle_skewness(const double* x, int n, double c, double v) { /* Stat.cpp:510-528 */
  if (!n || fabs(v) <= kEpsilon) return 0.0;
  double r = 0.0;
  int i = n;
  while (i--) { const double t = (x[i] - c) / v; r += t * t * t; }
  return r / n;
}
double afx_oracle_kurtosis(const double* x, int n, double c, double v) { /* Stat.cpp:532-554 */
  if (!n || fabs(v) <= kEpsilon) return 0.0;
  double r = 0.0;
  int i = n;
  while (i--) { const double t = (x[i] - c) / v; const double tt = t * t; r += tt * tt; }
  r /= n;
  r -= 3.0;
  return r;
}
double afx_oracle_flatness(const double* x, int n) { /* Stat.cpp:558-574 */
  const double am = afx_oracle_mean(x, n);
  const double gm = afx_oracle_geometric_mean(x, n);
  if (am == 0.0) return 0.0;
  return gm / am;
}
/* TAudioMath::LinToDb(double), R/Source/Core/AudioTypes/Export/AudioMath.inl:55-70;
 * MMinusInfInDb = -200.0f (AudioMath.h:17) */
double afx_oracle_lin_to_db(double v) {
  const double f = 20.0 / log(10.0);
  if (v == 1.0) return 0.0;
  if (v > kEpsilon) return log(v) * f;
  return -200.0;
}
double afx_oracle_flatness_db(const double* x, int n) { /* SFlatnessDb, SA.cpp:129-133 */
  const double d = afx_oracle_lin_to_db(afx_oracle_flatness(x, n)) / -60.0;
  return (d < 1.0) ? d : 1.0; /* MMin(a,b) = a < b ? a : b */
}
double afx_oracle_correlation(const double* a, const double* b, int n) { /* Stat.cpp:578-638 */
  if (!n) return 0.0;
  double ss1 = 0, ss2 = 0, ss11 = 0, ss12 = 0, ss22 = 0;
  for (int i = 0; i < n; ++i) {
    const double p = a[i], q = b[i];
    ss12 = ss12 + p * q; ss1 = ss1 + p; ss11 = ss11 + p * p; ss2 = ss2 + q; ss22 = ss22 + q * q;
  }
  ss1 = ss1 / n; ss2 = ss2 / n;
  {
    const double denom2 = (ss11 - ss1 * ss1 * n) * (ss22 - ss2 * ss2 * n);
    const double num = ss12 - (ss1 * ss2 * n);
    if (fabs(denom2) > kEpsilon) return num / sqrt(denom2);
  }
  return 0.0;
}

/* TStatistics::Calc, Stat.cpp:12-90 */
void afx_oracle_calc_statistics(const double* x, int n, double* o) {
  if (n > 1) {
    o[0] = afx_oracle_min(x, n); o[1] = afx_oracle_max(x, n);
    o[2] = afx_oracle_median(x, n); o[3] = afx_oracle_mean(x, n);
    o[4] = afx_oracle_geometric_mean(x, n); o[5] = afx_oracle_variance(x, n, o[3]);
    o[6] = afx_oracle_centroid(x, n); o[7] = afx_oracle_spread(x, n, o[6]);
    o[8] = afx_oracle_skewness(x, n, o[6], o[7]); o[9] = afx_oracle_kurtosis(x, n, o[6], o[7]);
    o[10] = (o[3] == 0.0) ? 0.0 : o[4] / o[3];
    if (n > 2) {
      double* d = (double*)malloc(sizeof(double) * (n - 1));
      for (int i = 0; i < n - 1; ++i) d[i] = fabs(x[i + 1] - x[i]);
      o[11] = afx_oracle_mean(d, n - 1);
      o[12] = afx_oracle_variance(d, n - 1, o[11]);
      free(d);
    } else { o[11] = 0.0; o[12] = 0.0; }
  } else if (n > 0) {
    o[0] = x[0]; o[1] = x[0]; o[3] = x[0]; o[5] = 0.0; o[11] = 0.0; o[12] = 0.0;
  } else {
    o[0] = 0.0; o[1] = 0.0; o[3] = 0.0; o[5] = 0.0; o[11] = 0.0; o[12] = 0.0;
  }
}

/* ---- LibXtract scalars -------------------------------------------------------------------- */

/* xtract_rms_amplitude, Xt/src/scalar.c:624-636 (reverse summation) */
static double xt_rms(const double* x, int n) {
  double r = 0.0;
  int i = n;
  while (i--) r += x[i] * x[i];
  return sqrt(r / (double)n);
}
/* xtract_rolloff, Xt/src/scalar.c:472-492 with argv = {44100/(2048/2) = 43 (int division,
 * SA.cpp:1892), 85.0f}; the forward loop has no bound on n, as in the reference */
static double xt_rolloff(const double* x, int n, double bin_width, double percentile) {
  double pivot = 0.0, temp = 0.0;
  int i = n;
  while (i--) pivot += x[i];
  pivot *= percentile / 100.0;
  for (i = 0; temp < pivot; ++i) temp += x[i];
  return i * bin_width;
}
/* xtract_mfcc + xtract_dct, Xt/src/vector.c:350-391; XTRACT_LOG_LIMIT = 2e-42
 * (Xt/src/xtract_macros_private.h:34-35) */
static void xt_mfcc(const afx_oracle* o, const double* mag, double* out) {
  const int N = o->fft / 2;
  double e[N_CEP];
  for (int f = 0; f < N_CEP; ++f) {
    const double* t = o->mel + (size_t)f * N;
    double r = 0.0;
    for (int k = 0; k < N; ++k) r += mag[k] * t[k];
    e[f] = log(r < 2e-42 ? 2e-42 : r);
  }
  for (int n = 0; n < N_CEP; ++n) {
    double t = 0.0;
    for (int m = 1; m <= N_CEP; ++m) t += e[m - 1] * cos(M_PI * (n / (double)N_CEP) * (m - 0.5));
    out[n] = t;
  }
}

static double nan_to_zero(double v) { return (v != v) ? 0.0 : v; }

/* ---- per-frame descriptors (SA.cpp:871-872, 946-972) --------------------------------------- */

static const double kBand28[N_BANDS] = { /* SA.cpp:2015-2019 */
  50.0, 100.0, 150.0, 200.0, 300.0, 400.0, 510.0, 630.0, 770.0, 920.0, 1080.0, 1270.0, 1480.0,
  1720.0, 2000.0, 2320.0, 2700.0, 3150.0, 3700.0, 4400.0, 5300.0, 6400.0, 7700.0, 9500.0, 12000.0,
  15500.0, 19000.0, 22050.0 };
static const double kBand14[N_SUB] = { /* SA.cpp:2077-2080 */
  50.0, 100.0, 200.0, 400.0, 630.0, 920.0, 1270.0, 1720.0, 2320.0, 3150.0, 4400.0, 6400.0, 9500.0,
  15500.0 };

static void frame_descriptors(const afx_oracle* o, const double* x, const double* mag,
                              const double* last, double* sorted_scratch, double* out) {
  const int half = o->fft / 2;
  const double* m = mag + o->first_bin;
  const int n = o->bin_count;
  const double fpb = (double)(o->sample_rate / o->fft);
  const int first = d2i_round(20.0 / fpb);
  int b, i;

  memcpy(out + AFXO_MAG, mag, sizeof(double) * half);

  /* CalcAmplitudePeak / CalcAmplitudeRms over the hop, SA.cpp:1760-1783 */
  {
    double pk = 0.0;
    for (i = 0; i < o->hop; ++i) { const double a = fabs(x[i]); if (a > pk) pk = a; }
    out[AFXO_AMP_PEAK] = pk;
    out[AFXO_AMP_RMS] = nan_to_zero(xt_rms(x, o->hop));
  }
  out[AFXO_SRMS] = nan_to_zero(xt_rms(m, n));                         /* SA.cpp:1808-1818 */
  {
    const double c = afx_oracle_centroid(m, n);                        /* SA.cpp:1822-1837 */
    const double v = afx_oracle_spread(m, n, c);
    out[AFXO_CENTROID] = c; out[AFXO_SPREAD] = v;
    out[AFXO_SKEW] = afx_oracle_skewness(m, n, c, v);                  /* SA.cpp:1858-1883 */
    out[AFXO_KURT] = afx_oracle_kurtosis(m, n, c, v);
  }
  out[AFXO_ROLLOFF] = nan_to_zero(                                     /* SA.cpp:1887-1901 */
      xt_rolloff(m, n, (double)(o->sample_rate / (o->fft / 2)), 85.0f));
  out[AFXO_FLATNESS] = nan_to_zero(afx_oracle_flatness_db(m, n));      /* SA.cpp:1905-1915 */
  out[AFXO_FLUX] = afx_oracle_correlation(m, last + o->first_bin, n);  /* SA.cpp:1919-1933 */

  /* CalcSpectralBandFeatures, SA.cpp:2067-2308 */
  {
    int nbins[N_SUB];
    int cur = first;
    double contrast_sum = 0.0;
    for (b = 0; b < N_SUB; ++b) {
      const int start = (b == 0) ? first : d2i_round(kBand14[b - 1] / fpb);
      const int end = d2i_round(kBand14[b] / fpb);
      nbins[b] = end - start + 1;
    }
    memcpy(sorted_scratch, mag, sizeof(double) * half);
    for (b = 0; b < N_SUB; ++b) {
      const int nb = (nbins[b] < half - cur) ? nbins[b] : (half - cur);
      double* s = sorted_scratch + cur;
      const double band_mean = afx_oracle_mean(s, nb);
      double rms = 0.0, thr = 0.0, cplx = 0.0, sum, valley, peak;
      int nn;
      for (i = 0; i < nb; ++i) rms += s[i] * s[i];
      rms = sqrt(rms / (double)nb);
      out[AFXO_SUB_RMS + b] = rms;
      out[AFXO_SUB_FLAT + b] = afx_oracle_flatness_db(s, nb);
      out[AFXO_SUB_FLUX + b] = afx_oracle_correlation(s, last + cur, nb);
      for (i = 0; i < nb; ++i) thr = (thr > s[i]) ? thr : s[i];
      thr *= 0.25; /* MPeakThreshold, SA.cpp:47 */
      if (thr > 0.0) {
        for (i = 0; i < nb; ++i) {
          const int k = cur + i; /* unsorted spectrum, may look into the neighbouring band */
          if (mag[k] > thr && k > 0 && k < half - 1 && mag[k] > mag[k - 1] && mag[k] > mag[k + 1])
            cplx += 1.0;
        }
      }
      out[AFXO_SUB_CPLX + b] = cplx;
      qsort(s, nb, sizeof(double), cmp_double); /* std::sort ascending, SA.cpp:2200-2201 */
      nn = (int)(0.3 * nb); if (nn < 1) nn = 1;
      sum = 0; for (i = 0; i < nn && i < nb; ++i) sum += s[i];
      valley = sum / nn + 1e-30;
      sum = 0; for (i = nb; i > nb - nn; --i) sum += s[i - 1];
      peak = sum / nn + 1e-30;
      out[AFXO_SUB_CONTRAST + b] = -1.0 * pow(peak / valley, 1.0 / log(band_mean + 1e-30));
      contrast_sum += out[AFXO_SUB_CONTRAST + b];
      cur += nb;
    }
    out[AFXO_CONTRAST] = contrast_sum / N_SUB; /* SA.cpp:2252-2260 */
  }

  /* CalcSpectrumBands, SA.cpp:2007-2048 */
  for (b = 0; b < N_BANDS; ++b) out[AFXO_BANDS + b] = 0.0;
  for (b = 0; b < N_BANDS; ++b) {
    int start = d2i_round((b == 0) ? (double)first : kBand28[b - 1] / fpb);
    int end;
    if (start >= half) break;
    end = d2i_round(kBand28[b] / fpb);
    if (half < end) end = half;
    for (i = start; i < end; ++i) out[AFXO_BANDS + b] += mag[i] * mag[i];
  }

  xt_mfcc(o, mag, out + AFXO_MFCC); /* CalcCepstrumBands, SA.cpp:2052-2063 */
}

int afx_oracle_sample_rate(const afx_oracle* o) { return o->sample_rate; }

int64_t afx_oracle_analysed_length(const afx_oracle* o, int64_t n_samples, int apply_cap) {
  int64_t len = n_samples;
  if (apply_cap) { /* MAnalyzationDurationMaxInMs = 1000*20, SA.cpp:37, 760-764 */
    const int64_t cap = ms_to_samples(o->sample_rate, 1000 * 20);
    if (cap < len) len = cap;
  }
  return len;
}

int64_t afx_oracle_num_frames(const afx_oracle* o, int64_t n_samples, int apply_cap) {
  const int64_t len = afx_oracle_analysed_length(o, n_samples, apply_cap);
  int64_t f = 0, n;
  for (n = 0; (n + o->fft - 1) < len; n += o->hop) ++f; /* SA.cpp:814 */
  return f;
}

int64_t afx_oracle_run(const afx_oracle* o, const double* x, int64_t n_samples, int apply_cap,
                       double* records) {
  const int64_t frames = afx_oracle_num_frames(o, n_samples, apply_cap);
  const int n = o->fft;
  double* re = (double*)malloc(sizeof(double) * n * 5);
  double* im = re + n; double* mag = im + n; double* last = mag + n; double* scratch = last + n;
  int64_t f;
  memset(last, 0, sizeof(double) * n);
  for (f = 0; f < frames; ++f) {
    const double* fx = x + f * o->hop;
    stft_frame(o, fx, re, im, mag);
    if (f == 0) memcpy(last, mag, sizeof(double) * n);  /* SA.cpp:937-940 */
    frame_descriptors(o, fx, mag, last, scratch, records + (size_t)f * AFXO_RECORD);
    memcpy(last, mag, sizeof(double) * n);              /* SA.cpp:975 */
  }
  free(re);
  return frames;
}

int64_t afx_oracle_run_mfcc(const afx_oracle* o, const double* x, int64_t n_samples, double* mfcc) {
  const int64_t frames = afx_oracle_num_frames(o, n_samples, 0);
  const int n = o->fft;
  double* re = (double*)malloc(sizeof(double) * n * 3);
  double* im = re + n; double* mag = im + n;
  int64_t f;
  for (f = 0; f < frames; ++f) {
    stft_frame(o, x + f * o->hop, re, im, mag);
    xt_mfcc(o, mag, mfcc + (size_t)f * N_CEP);
  }
  free(re);
  return frames;
}

/* ---- stateful neighbours ---------------------------------------------------- */

/* TStatistics::Peaks, Stat.cpp:140-232: boundary peaks, interior peaks after a strict climb, plateaus
 * report their middle bin and their first value, everything strictly above the threshold. */
int afx_oracle_peaks(const double* a, int n, double thr, int* bins, double* vals) {
  int count = 0, i = 0, j;
  if (n <= 2) return 0;
  if (a[0] > a[1] && a[0] > thr) { bins[count] = 0; vals[count++] = a[0]; }
  for (;;) {
    while (i + 1 < n - 1 && a[i] >= a[i + 1]) ++i;
    while (i + 1 < n - 1 && a[i] < a[i + 1]) ++i;
    j = i;
    while (j + 1 < n - 1 && a[j] == a[j + 1]) ++j;
    if (j + 1 < n - 1 && a[j + 1] < a[j] && a[j] > thr) {
      bins[count] = (j != i) ? (i + j) / 2 : j;
      vals[count++] = (j != i) ? a[i] : a[j];
    }
    i = j;
    if (i + 1 >= n - 1) {
      if (i == n - 2 && a[i - 1] < a[i] && a[i + 1] < a[i] && a[i] > thr) { bins[count] = i; vals[count++] = a[i]; }
      break;
    }
  }
  if (a[n - 1] > a[n - 2] && a[n - 1] > thr) { bins[count] = n - 1; vals[count++] = a[n - 1]; }
  return count;
}

/* aubio_silence_detection, Aubio/Dist/src/mathutils.c:345-357, 605-615 */
static int au_silent(const double* x, int n, double threshold_db) {
  double e = 0.0;
  int i;
  for (i = 0; i < n; ++i) e += x[i] * x[i];
  return 10.0 * log10(e / n) < threshold_db;
}

/* CalcAmplitudeEnvelope, SA.cpp:1787-1804; TEnvelopeDetector kFast, Envelopes.cpp:55-70, Envelopes.inl:14-18
 * (MUnDenormalize is empty on x64, InlineMath.h:36-53) */
static double hop_envelope(const afx_oracle* o, const double* x) {
  const double coef = pow(0.01, 1000.0 / (8.0 * (double)o->sample_rate));
  double env = 0.0, top = 0.0;
  int i;
  for (i = 0; i < o->hop; ++i) {
    const double in = fabs(x[i]);
    env = in + coef * (env - in);
    if (env > top) top = env;
  }
  return top;
}

/* aubio_pitchyinfast_do, Aubio/Dist/src/pitch/pitchyinfast.c:81-170, with r_t(tau) = sum_{j<W} x[j] x[j+tau]
 * summed directly instead of through aubio's FFT; then aubio_pitch_do_yinfast + aubio_pitch_do,
 * pitch.c:399-406, 450-462.  yin[] is W = fft/2 long; returns the frequency.
 * QUIRK (part of the reference's results, reproduced): with aubio's Ooura back end -- the one the
 * reference builds on Linux (Aubio/Linux/build.sh: no FFTW, no IPP) and on macOS (Aubio/Mac/config.h) --
 * aubio_fft_rdo_complex rescales the inverse rdft by 1/N where Ooura needs 2/N (spectral/fft.c:464-476),
 * so the correlation comes back exactly halved and the "difference function" is
 * sqdiff(tau) - r_t(tau), not sqdiff(tau) - 2 r_t(tau). */
static double yinfast_pitch(const afx_oracle* o, const double* x, double tol, double* yin, double* confidence) {
  const int B = o->fft, W = B / 2;
  int tau, j, period_i = -1;
  double s0 = 0.0, run, tmp2 = 0.0, period;
  for (j = 0; j < W; ++j) s0 += x[j] * x[j];
  run = s0;
  for (tau = 0; tau < W; ++tau) {
    double r = 0.0;
    if (tau > 0) { run -= x[tau - 1] * x[tau - 1]; run += x[W + tau - 1] * x[W + tau - 1]; }
    for (j = 0; j < W; ++j) r += x[j] * x[j + tau];
    yin[tau] = (run + s0) - 2.0 * (0.5 * r);
  }
  yin[0] = 1.0;
  for (tau = 1; tau < W; ++tau) {
    tmp2 += yin[tau];
    if (tmp2 != 0) yin[tau] *= tau / tmp2; else yin[tau] = 1.0;
    if (tau > 4 && yin[tau - 3] < tol && yin[tau - 3] < yin[tau - 2]) { period_i = tau - 3; break; }
  }
  if (period_i < 0) { /* fvec_min_elem, mathutils.c:250-265: the last of equal minima */
    double m = yin[0];
    period_i = 0;
    for (j = 0; j < W; ++j) if (!(m < yin[j])) { period_i = j; m = yin[j]; }
  }
  /* fvec_quadratic_peak_pos, mathutils.c:494-506 */
  if (period_i == 0 || period_i == W - 1) period = period_i;
  else {
    const double a = yin[period_i - 1], b = yin[period_i], c = yin[period_i + 1];
    period = period_i + 0.5 * (a - c) / (a - 2.0 * b + c);
  }
  *confidence = 1.0 - yin[(unsigned)period];   /* peak_pos is a uint_t, pitchyinfast.c:38, 160-169 */
  {
    double f = (period > 0) ? o->sample_rate / (period + 0.) : 0.0;
    if (au_silent(x, B, -48.0)) f = 0.0;
    return f;
  }
}

/* CalcAutoCorrelation, SA.cpp:2312-2398; TAutocorrelation::Calc, Autocorrelation.cpp:62-106 */
static double auto_correlation(const afx_oracle* o, const double* x, int64_t remaining_in) {
  const int min_period = ms_to_samples(o->sample_rate, 0.8f);
  const int seek_width = ms_to_samples(o->sample_rate, 12.0f);
  const int max_seek = o->fft / 2;
  int remaining = (int)remaining_in, i, j, seek_off, period, width;
  const double* start = x; const double* end;
  double best = 0.0, r0 = 0.0;
  for (i = 0; i < (remaining < max_seek ? remaining : max_seek) - 1; ++i)
    if (x[i + 1] > x[i]) { start = x + i; remaining -= i; break; }
  seek_off = remaining < min_period ? remaining : min_period;
  end = start + seek_off;
  for (i = 0; i < ((remaining - seek_off) < max_seek ? (remaining - seek_off) : max_seek) - 1; ++i)
    if (start[seek_off + i + 1] > start[seek_off + i]) { end = start + seek_off + i; break; }
  period = (int)(end - start);
  if (!remaining || period >= remaining) return 0.0;
  width = remaining < seek_width ? remaining : seek_width;
  for (j = 0; j < width; ++j) r0 += start[j] * start[j];
  for (i = period / 2; i < width; ++i) {
    double r = 0.0;
    for (j = 0; j < width - i; ++j) r += start[j] * start[j + i];
    if (r0 != 0) r /= r0;
    if (r > best) best = r;
  }
  return best;
}

int64_t afx_oracle_run_neighbours(const afx_oracle* o, const double* x, int64_t n_samples, int apply_cap,
                                  double* records) {
  const int64_t frames = afx_oracle_num_frames(o, n_samples, apply_cap);
  const int n = o->fft, half = n / 2;
  double* re = (double*)malloc(sizeof(double) * n * 8);
  double* im = re + n; double* mag = im + n; double* wh = mag + n; double* pk = wh + n;
  double* follow = pk + n; double* yin = follow + n; double* pval = yin + n;
  int* pbin = (int*)malloc(sizeof(int) * n);
  /* new_aubio_spectral_whitening + set_relax_time(22), awhitening.c:53-87, SA.cpp:805-809:
   * r_decay = pow(0.001, (hop / (float)rate) / relax), floor 1e-4, followers start at the floor */
  const double decay = pow(0.001, (double)((float)o->hop / (float)o->sample_rate) / 22.0);
  const double floor_v = 1.e-4;
  int64_t f;
  int i;
  for (i = 0; i <= half; ++i) follow[i] = floor_v;
  for (f = 0; f < frames; ++f) {
    const double* fx = x + f * o->hop;
    double* out = records + (size_t)f * AFXN_RECORD;
    double mx = 0.0, f0, conf, safe = 0.0;
    int npk, silent, count = 0;
    stft_frame(o, fx, re, im, mag);
    /* aubio_spectral_whitening_do, awhitening.c:41-51 (bins 0..fft/2, the last one is the cleared upper half) */
    for (i = 0; i <= half; ++i) {
      double t = decay * follow[i];
      if (t < floor_v) t = floor_v;
      follow[i] = mag[i] > t ? mag[i] : t;
      wh[i] = mag[i] / follow[i];
    }
    memcpy(out + AFXN_WHITE, wh, sizeof(double) * half);
    /* SCreatePeakSpectrum, SA.cpp:95-123, threshold MPeakThreshold = 0.25 of the maximum */
    mx = afx_oracle_max(wh, half);
    npk = afx_oracle_peaks(wh, half, 0.25 * mx, pbin, pval);
    memset(pk, 0, sizeof(double) * n);
    for (i = 0; i < npk; ++i) pk[pbin[i]] = pval[i];
    silent = au_silent(fx, o->hop, -48.0);               /* SA.cpp:865-868 */
    out[AFXN_SILENCE] = silent ? 1.0 : 0.0;
    out[AFXN_ENVELOPE] = hop_envelope(o, fx);
    /* SA.cpp:876-917 */
    f0 = yinfast_pitch(o, fx, 0.75, yin, &conf);
    conf = conf / 0.25;
    conf = conf < 0.0 ? 0.0 : (conf > 1.0 ? 1.0 : conf);
    if (f0 > 0.0 && conf > 0.2) safe = f0;
    else if (!silent) {
      const double cb = afx_oracle_centroid(mag, half);
      safe = (double)o->sample_rate / (double)o->fft * (cb > 0.0 ? cb : 0.0);
    }
    out[AFXN_F0] = f0; out[AFXN_F0_CONF] = conf; out[AFXN_F0_FAILSAFE] = safe;
    out[AFXN_AUTOCORR] = auto_correlation(o, fx, n_samples - f * o->hop);   /* SA.cpp:943-944 */
    /* CalcSpectralComplexity, SA.cpp:1937-1947: non-zero bins of the peak spectrum in the analysis range */
    for (i = 0; i < o->bin_count; ++i) count += pk[o->first_bin + i] != 0.0;
    out[AFXN_COMPLEXITY] = count;
    /* xtract_spectral_inharmonicity / xtract_harmonic_spectrum / xtract_tristimulus_* read their
     * frequencies from the upper half of the fft-sized buffer (Xt/src/scalar.c:304-326, 638-661,
     * vector.c:540-579), which SA.cpp:844-845 clears: every partial sits at 0 Hz, the harmonic spectrum
     * is empty, and all four descriptors are 0 (0/0 -> NaN -> 0 for inharmonicity, SA.cpp:1963-1964). */
    out[AFXN_INHARM] = out[AFXN_TRI1] = out[AFXN_TRI2] = out[AFXN_TRI3] = 0.0;
  }
  free(re); free(pbin);
  return frames;
}

/* CalcEffectiveLength, SA.cpp:1715-1755; TAudioMath::DbToLin(double) = exp(db ln10 / 20), SamplesToMs in float
 * (AudioMath.inl:108-123, 134-137) */
void afx_oracle_effective_length(const afx_oracle* o, const double* x, int64_t n_samples, double* out3) {
  static const double db[3] = { -48.0, -24.0, -12.0 };
  const int n = (int)n_samples;
  int s, f;
  for (s = 0; s < 3; ++s) {
    const double floor_v = exp(db[s] * (log(10.0) / 20.0));
    int lead = 0, trail = 0;
    for (f = 0; f < n; ++f, ++lead) if (fabs(x[f]) > floor_v) break;
    for (f = n - 1; f > lead; --f, ++trail) if (fabs(x[f]) > floor_v) break;
    out3[s] = (double)((float)(n - lead - trail) / ((float)o->sample_rate / 1000.0f)) / 1000.0;
  }
}

/* ---- LoadSample (SA:484-718), decoded interleaved PCM in, normalised mono double out ---- */

static float to_16bit_float(const void* pcm, int format, int64_t idx) {
  if (format == 0) return (float)((const int16_t*)pcm)[idx];           /* SampleConverter.h:446-449 */
  if (format == 1) {                                                    /* SampleConverter.h:474-486 */
    const unsigned char* b = (const unsigned char*)pcm + 3 * idx;
    const int32_t v = (int32_t)(((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16)) << 8);
    return (float)(v * 32768.0 / 2147483648.0);
  }
  if (format == 3) {                                                    /* SampleConverter.h:514-518 */
    const double dv = (double)((const int32_t*)pcm)[idx] * 32768.0 / 2147483648.0;
    const float f = (float)dv;
    return f < -32768.0f ? -32768.0f : (f > 32767.0f ? 32767.0f : f);
  }
  {                                                                     /* SampleConverter.h:529-533 (float32 / float64) */
    const double d = (format == 4 ? ((const double*)pcm)[idx] : (double)((const float*)pcm)[idx]) * 32768.0;
    return (float)(d < -32768.0 ? -32768.0 : (d > 32767.0 ? 32767.0 : d));
  }
}

double* afx_oracle_load_sample(const void* pcm, int format, int channels, int64_t n_frames, int fft_size,
                               afx_oracle_load_info* info) {
  return afx_oracle_load_sample_at(pcm, format, channels, n_frames, 44100, 44100, fft_size, info);
}

double* afx_oracle_load_sample_at(const void* pcm, int format, int channels, int64_t n_frames, int file_rate,
                                  int analyser_rate, int fft_size, afx_oracle_load_info* info) {
  const float scale = 65536 / 2.0f;   /* sScaleFactor = M16BitSampleRange / 2.0f, SA:533 */
  float* mono = (float*)malloc(sizeof(float) * (size_t)(n_frames > 0 ? n_frames : 1));
  int64_t n, lead, trail, audible, start_pad, end_pad, size;
  double rms = 0.0, max_amp, amplification, floor_lin, final_scaling;
  float mn, mx;
  double* out;
  /* mix down to mono in float, first channel as destination (SA:535-556) */
  for (n = 0; n < n_frames; ++n) {
    float d = to_16bit_float(pcm, format, n * channels);
    if (channels > 1) {
      const float mix = 1.0f / (float)channels;
      int c;
      for (c = 1; c < channels; ++c) d += to_16bit_float(pcm, format, n * channels + c);
      d *= mix;
    }
    mono[n] = d;
  }
  /* resample when the file is not at the analyser's rate (SA:563-607) */
  if ((double)file_rate / (double)analyser_rate != 1.0) {
    int64_t n_out = 0, n_written = 0;
    float* resampled = afx_oracle_resample(mono, n_frames, file_rate, analyser_rate, &n_out, &n_written);
    free(mono);
    mono = resampled;
    n_frames = n_out;
  }
  /* rms (SA:612-619) */
  for (n = 0; n < n_frames; ++n) { const double t = (double)(mono[n] / scale); rms += t * t; }
  info->rms_value = (float)fmin(1.0, sqrt(rms / (double)(1 * n_frames)));
  /* peak and normalisation factor (SA:624-637) */
  mn = mx = mono[0];
  for (n = 1; n < n_frames; ++n) { if (mono[n] < mn) mn = mono[n]; if (mono[n] > mx) mx = mono[n]; }
  max_amp = (double)(fabsf(mn) > fabsf(mx) ? fabsf(mn) : fabsf(mx));
  info->peak_value = (float)fmin(1.0, max_amp / scale);
  amplification = (max_amp > (double)1e-12f) ? scale / max_amp : 1.0;
  /* leading / trailing silence below -48 dB of full scale (SA:648-669) */
  floor_lin = scale * exp(-48.0 * (log(10.0) / 20.0));
  lead = 0;
  for (n = 0; n < n_frames; ++n, ++lead) if (fabs(amplification * mono[n]) > floor_lin) break;
  trail = 0;
  for (n = n_frames - 1; n > lead; --n, ++trail) if (fabs(amplification * mono[n]) > floor_lin) break;
  /* pad so that at least half of the last frame and one full frame get analysed (SA:681-696) */
  audible = n_frames - lead - trail;
  end_pad = ((audible % fft_size) < fft_size / 2) ? fft_size / 2 : 0;
  start_pad = (audible + end_pad < fft_size) ? fft_size - audible - end_pad : 0;
  size = audible + start_pad + end_pad;
  out = (double*)calloc((size_t)size, sizeof(double));
  final_scaling = amplification / scale;
  for (n = 0; n < audible; ++n) out[n + start_pad] = mono[n + lead] * final_scaling;   /* SA:712-718 */
  info->data_offset = (int32_t)(-lead + start_pad);
  info->silent_leading = (int32_t)lead;
  info->silent_trailing = (int32_t)trail;
  info->n_samples = size;
  free(mono);
  return out;
}

void afx_oracle_free(void* p) { free(p); }
