// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the *reference's own* compiled objects (LibXtract C sources, OouraFFT8g.cpp,
// AudioMath.cpp, Statistics.cpp, built where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/) through the per-frame loop of
// TSampleAnalyser::AnalyzeLowLevelDescriptors (SampleAnalyser.cpp:814-976) for the
// descriptors in SURVEY.md section 8(a).  The Calc* member bodies are private and
// SampleAnalyser.cpp itself does not build here (aubio/Shark/LightGBM/CoreTypes),
// so this driver restates only their *argument slicing* and calls the real
// xtract_*/ooura_cdft/TAudioMath/TStatistics functions for every piece of arithmetic
// that exists as a linkable reference function.  Bodies that are inline in
// SampleAnalyser.cpp (band loops, contrast) are restated and marked as such.
//
// Usage:
//   ref_driver tables  <out.bin>                 window[2048] + mel[14][1024] doubles
//   ref_driver frames  <in.bin> <out.bin> [cap]  per-frame records (see kRecord)
//   ref_driver neighbours <in.bin> <out.bin> [cap]  per-frame records of the stateful neighbours (kNeigh)
//   ref_driver load    <in.bin> <out.bin>        LoadSample normalisation front end (SampleAnalyser.cpp:484-718)
//   ref_driver efflen  <in.bin> <out.bin>        effective lengths at -48/-24/-12 dB per buffer (3 doubles each)
//   ref_driver msgpack <in.bin> <out.bin>        BLOB encoding of a VR / VVR column (SqliteSampleDescriptorPool.cpp:596-713)
//   ref_driver onsetfft <in.bin> <out.bin>       onset STFT front end of the rhythm tracker (OnsetDetector.cpp:116-160)
//   ref_driver beattrack <in.bin> <out.bin>      aubio beat tracking pass of TRhythmTracker::CalculateTempo (bpm, confidence)
//   ref_driver time    <n_frames> <seed>         C2 subset timing (STFT + MFCC), prints frames/s
//
// in.bin : int64 n_bufs ; per buffer: int64 n_samples, double[n_samples]
// out.bin: int64 n_frames ; double[n_frames][kRecord]

#include "libresample.h"   // 3rdParty/Resample/Dist/include: the reference's libresample 0.1.3
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <chrono>
#include <random>

extern "C" {
#include "xtract/libxtract.h"
}
#include "AudioTypes/Source/OouraFFT8g.h"
#include "AudioTypes/Export/AudioMath.h"
#include "FeatureExtraction/Export/Statistics.h"
#include "CoreTypes/Export/Array.h"
#include "CoreFileFormats/Export/SampleConverter.h"
#include "AudioTypes/Export/AudioTypes.h"
#include "AudioTypes/Export/Envelopes.h"
#include "FeatureExtraction/Source/Autocorrelation.h"
#include "AudioTypes/Export/Fourier.h"
#include <msgpack.hpp>   // 3rdParty/Msgpack/Dist/include (header-only, version 2.1), as SqliteSampleDescriptorPool.cpp uses it
extern "C" {
#include "aubio.h"   // 3rdParty/Aubio/Dist/src, smpl_t = double (HAVE_AUBIO_DOUBLE, as Export/Aubio.h sets it)
#include "tempo/beattracking.h"   // as 3rdParty/Aubio/Export/Aubio.h:21 includes it
}

static const int kSampleRate = 44100, kFft = 2048, kHop = 1024;
static const int kNumCep = 14, kNumBands = 28, kNumSub = 14;
// record layout (doubles)
enum {
  oMag = 0, oMfcc = oMag + 1024, oSRms = oMfcc + 14, oCentroid, oSpread, oSkew, oKurt,
  oRolloff, oFlatness, oFlux, oBands = oFlux + 1, oSubRms = oBands + 28,
  oSubFlat = oSubRms + 14, oSubFlux = oSubFlat + 14, oSubCplx = oSubFlux + 14,
  oSubContrast = oSubCplx + 14, oContrast = oSubContrast + 14, oAmpPeak, oAmpRms,
  kRecord
};

struct TRef {
  int mFirstBin, mLastBin, mBinCount;
  double* mpWindow;
  xtract_mel_filter mMel;
  // TFftTransformComplex state (Fourier.cpp:97-108): ooura work areas
  std::vector<double> mInterleaved, mW, mRe, mIm;
  std::vector<int> mIp;

  TRef() : mInterleaved(2 * kFft), mW(kFft), mRe(kFft), mIm(kFft), mIp(kFft) {
    // SampleAnalyser.cpp:171-175 (note the integer division)
    const double FrequenciesPerBin = kSampleRate / kFft;
    mFirstBin = TMath::d2iRound(20.0 / FrequenciesPerBin);
    mLastBin = TMath::d2iRound(15500.0 / FrequenciesPerBin);
    mBinCount = mLastBin - mFirstBin + 1;
    // SampleAnalyser.cpp:178-181
    mpWindow = xtract_init_window(kFft, XTRACT_HANN);
    for (int i = 0; i < kFft; ++i) mpWindow[i] *= 2.0;
    // SampleAnalyser.cpp:184-197
    mMel.n_filters = kNumCep;
    mMel.filters = (double**)malloc(kNumCep * sizeof(double*));
    for (int k = 0; k < kNumCep; ++k) mMel.filters[k] = (double*)malloc((kFft / 2) * sizeof(double));
    xtract_init_mfcc(kFft / 2, kSampleRate / 2, XTRACT_EQUAL_GAIN, 20.0, 15500.0, kNumCep, mMel.filters);
    mIp[0] = 0;
  }

  // TFftTransformComplex::ForwardInplace, generic branch (Fourier.cpp:243-270)
  void Forward() {
    for (int i = 0; i < kFft; ++i) { mInterleaved[2*i] = mRe[i]; mInterleaved[2*i+1] = mIm[i]; }
    mIp[0] = 0;
    ooura_cdft(2 * kFft, 1, mInterleaved.data(), mIp.data(), mW.data());
    for (int i = 0; i < kFft; ++i) { mRe[i] = mInterleaved[2*i]; mIm[i] = mInterleaved[2*i+1]; }
    const double ScaleFactor = 1.0f / kFft;
    for (int i = 0; i < kFft; ++i) { mRe[i] *= ScaleFactor; mIm[i] *= ScaleFactor; }
  }

  // SampleAnalyser.cpp:826-845
  void Stft(const double* frame, double* mag /*[2048]*/) {
    std::vector<double> windowed(kFft);
    xtract_windowed(frame, kFft, mpWindow, windowed.data());
    memcpy(mRe.data(), windowed.data(), kFft * sizeof(double));
    memset(mIm.data(), 0, kFft * sizeof(double));
    Forward();
    TAudioMath::Magnitude(mRe.data(), mIm.data(), mag, kFft / 2);
    memset(mag + kFft / 2, 0, (kFft / 2) * sizeof(double));
  }
};

// SampleAnalyser.cpp:129-133
static double SFlatnessDb(const double* pX, int Length) {
  const double Flatness = TStatistics::Flatness(pX, Length);
  return MMin(TAudioMath::LinToDb(Flatness) / -60.0, 1.0);
}

static double NanToZero(double v) { return (v != v) ? 0.0 : v; }

static void Frame(TRef& R, const double* x, const double* mag, const double* last, double* out) {
  memcpy(out + oMag, mag, 1024 * sizeof(double));
  const double* m = mag + R.mFirstBin;
  const int n = R.mBinCount;
  // CalcAmplitudePeak / CalcAmplitudeRms on the hop (SampleAnalyser.cpp:871-872, 1760-1783)
  {
    double pk = 0.0;
    for (int i = 0; i < kHop; ++i) pk = std::max(pk, std::fabs(x[i]));
    out[oAmpPeak] = pk;
    double rms = 0.0; xtract_rms_amplitude(x, kHop, NULL, &rms);
    out[oAmpRms] = NanToZero(rms);
  }
  // CalcSpectralRms (1808-1818)
  { double rms = 0.0; xtract_rms_amplitude(m, n, NULL, &rms); out[oSRms] = NanToZero(rms); }
  // CalcSpectralCentroidAndSpread (1822-1837), SkewnessAndKurtosis (1858-1883)
  const double C = TStatistics::Centroid(m, n);
  const double S = TStatistics::Spread(m, n, C);
  out[oCentroid] = C; out[oSpread] = S;
  out[oSkew] = TStatistics::Skewness(m, n, C, S);
  out[oKurt] = TStatistics::Kurtosis(m, n, C, S);
  // CalcSpectralRolloff (1887-1901)
  {
    double Arguments[4] = { 0 };
    Arguments[0] = kSampleRate / (kFft / 2);
    Arguments[1] = 85.0f;
    double Rolloff = 0.0; xtract_rolloff(m, n, Arguments, &Rolloff);
    out[oRolloff] = NanToZero(Rolloff);
  }
  // CalcSpectralFlatness (1905-1915)
  out[oFlatness] = NanToZero(SFlatnessDb(m, n));
  // CalcSpectralFlux (1919-1933)
  out[oFlux] = TStatistics::Flux(m, last + R.mFirstBin, n);

  // CalcSpectralBandFeatures (SampleAnalyser.cpp:2067-2308): the body is inline member code, so its
  // control flow is restated here; Mean / Flux / Flatness go through the reference's TStatistics.
  {
    static const double edges_hz[kNumSub] = { 50.0, 100.0, 200.0, 400.0, 630.0, 920.0, 1270.0,
      1720.0, 2320.0, 3150.0, 4400.0, 6400.0, 9500.0, 15500.0 };
    const double hz_per_bin = kSampleRate / kFft;                    // integer division, as there
    const int first_bin = TMath::d2iRound(20.0 / hz_per_bin);
    int width[kNumSub];
    for (int b = 0; b < kNumSub; ++b) {
      const int lo = (b == 0) ? first_bin : TMath::d2iRound(edges_hz[b - 1] / hz_per_bin);
      width[b] = TMath::d2iRound(edges_hz[b] / hz_per_bin) - lo + 1;
    }
    std::vector<double> cur(mag, mag + kFft / 2), prv(last, last + kFft / 2);  // cur gets sorted band by band
    double contrast_total = 0.0;
    int at = first_bin;
    for (int b = 0; b < kNumSub; ++b) {
      const int nb = std::min(width[b], (int)cur.size() - at);
      double* seg = cur.data() + at;
      const double seg_mean = TStatistics::Mean(seg, nb);
      double sq = 0.0, top = 0.0;
      for (int i = 0; i < nb; ++i) { sq += seg[i] * seg[i]; top = std::max(top, seg[i]); }
      out[oSubRms + b] = ::sqrt(sq / (double)nb);
      out[oSubFlat + b] = SFlatnessDb(seg, nb);
      out[oSubFlux + b] = TStatistics::Flux(seg, prv.data() + at, nb);
      // strict local maxima of the *unsorted* spectrum above a quarter of the band maximum
      const double thr = 0.25 * top;
      double peaks = 0;
      if (thr > 0.0)
        for (int k = at; k < at + nb; ++k)
          if (mag[k] > thr && k > 0 && k < (int)cur.size() - 1 && mag[k] > mag[k - 1] && mag[k] > mag[k + 1]) ++peaks;
      out[oSubCplx + b] = peaks;
      // contrast from the mean of the lowest / highest 30 % of the sorted band
      std::sort(seg, seg + nb);
      const int take = std::max(1, (int)(0.3 * nb));
      double lo_sum = 0, hi_sum = 0;
      for (int i = 0; i < take && i < nb; ++i) lo_sum += seg[i];
      for (int i = nb; i > nb - take; --i) hi_sum += seg[i - 1];
      const double valley = lo_sum / take + 1e-30, peak = hi_sum / take + 1e-30;
      out[oSubContrast + b] = -1.0 * ::pow(peak / valley, 1.0 / ::log(seg_mean + 1e-30));
      contrast_total += out[oSubContrast + b];
      at += nb;
    }
    out[oContrast] = contrast_total / kNumSub;
  }

  // CalcSpectrumBands (SampleAnalyser.cpp:2007-2048): inline body, restated
  {
    static const double edges_hz[kNumBands] = { 50.0, 100.0, 150.0, 200.0, 300.0, 400.0, 510.0, 630.0,
      770.0, 920.0, 1080.0, 1270.0, 1480.0, 1720.0, 2000.0, 2320.0, 2700.0, 3150.0, 3700.0, 4400.0,
      5300.0, 6400.0, 7700.0, 9500.0, 12000.0, 15500.0, 19000.0, 22050.0 };
    const double hz_per_bin = kSampleRate / kFft;
    const int first_bin = TMath::d2iRound(20.0 / hz_per_bin);
    for (int b = 0; b < kNumBands; ++b) out[oBands + b] = 0.0;
    for (int b = 0; b < kNumBands; ++b) {
      const int lo = TMath::d2iRound((b == 0) ? (double)first_bin : edges_hz[b - 1] / hz_per_bin);
      if (lo >= kFft / 2) break;
      const int hi = std::min(kFft / 2, TMath::d2iRound(edges_hz[b] / hz_per_bin));
      for (int k = lo; k < hi; ++k) out[oBands + b] += mag[k] * mag[k];
    }
  }

  // CalcCepstrumBands (2052-2063)
  xtract_mfcc(mag, kFft / 2, &R.mMel, out + oMfcc);
}

static int CmdTables(const char* path) {
  TRef R;
  FILE* f = fopen(path, "wb"); if (!f) return 1;
  fwrite(R.mpWindow, sizeof(double), kFft, f);
  for (int k = 0; k < kNumCep; ++k) fwrite(R.mMel.filters[k], sizeof(double), kFft / 2, f);
  fclose(f);
  printf("first=%d last=%d count=%d\n", R.mFirstBin, R.mLastBin, R.mBinCount);
  return 0;
}

static int CmdFrames(const char* in, const char* outp, bool cap) {
  TRef R;
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int64_t nb = 0; if (fread(&nb, 8, 1, fi) != 1) return 1;
  std::vector<double> recs;
  int64_t total = 0;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t ns = 0; if (fread(&ns, 8, 1, fi) != 1) return 1;
    std::vector<double> x((size_t)ns);
    if (ns && fread(x.data(), 8, (size_t)ns, fi) != (size_t)ns) return 1;
    // SampleAnalyser.cpp:760-764, 814
    int64_t len = ns;
    if (cap) len = std::min<int64_t>(len, TAudioMath::MsToSamples(kSampleRate, 1000 * 20));
    std::vector<double> mag(kFft, 0.0), last(kFft, 0.0);
    for (int64_t n = 0; (n + kFft - 1) < len; n += kHop) {
      R.Stft(x.data() + n, mag.data());
      if (n == 0) last = mag;                    // SampleAnalyser.cpp:937-940
      recs.resize((size_t)(total + 1) * kRecord);
      Frame(R, x.data() + n, mag.data(), last.data(), recs.data() + (size_t)total * kRecord);
      last = mag;                                // SampleAnalyser.cpp:975
      ++total;
    }
  }
  fclose(fi);
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  fwrite(&total, 8, 1, fo);
  fwrite(recs.data(), 8, recs.size(), fo);
  fclose(fo);
  return 0;
}

// ---- the stateful neighbours of the loop: SampleAnalyser.cpp:849-927, 942-964 ----
// record: silence, envelope, f0, f0 confidence, fail-safe f0, autocorrelation, spectral complexity,
// inharmonicity, tristimulus 1..3, then the whitened spectrum [1024] (stage debugging)
enum { nSilence = 0, nEnvelope, nF0, nF0Conf, nF0FailSafe, nAutoCorr, nComplexity, nInharm, nTri1, nTri2, nTri3,
       nWhite, kNeigh = nWhite + 1024 };

// TStatistics::Peaks (Statistics.cpp:140-232) returns a TList, which needs CoreTypes' allocator and does not
// link here; its scan is restated (pinned by the reference's own vector, TestStatistics.cpp:16-33, in
// `ref_driver peakstest`).  Writes value at the reported bin for every peak above the threshold.
static int PeakScan(const double* a, int n, double thr, int* bins, double* vals) {
  int count = 0;
  if (n <= 2) return 0;
  if (a[0] > a[1] && a[0] > thr) { bins[count] = 0; vals[count++] = a[0]; }
  int i = 0;
  for (;;) {
    while (i + 1 < n - 1 && a[i] >= a[i + 1]) ++i;      // descend
    while (i + 1 < n - 1 && a[i] < a[i + 1]) ++i;       // climb
    int j = i;
    while (j + 1 < n - 1 && a[j] == a[j + 1]) ++j;      // plateau
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

static int CmdPeaksTest() {
  const double seq[] = { 1, 2, 2, 2, 0, 5, 6 };
  int bins[8]; double vals[8];
  for (double thr : { 0.0, 2.0 }) {
    const int c = PeakScan(seq, 7, thr, bins, vals);
    printf("thr=%g:", thr);
    for (int i = 0; i < c; ++i) printf(" (%d,%g)", bins[i], vals[i]);
    printf("\n");
  }
  return 0;
}

// CalcAutoCorrelation (SampleAnalyser.cpp:2312-2398): member body restated, TAutocorrelation::Calc is the
// reference's object
static double AutoCorrelation(const double* x, int remaining) {
  const int min_period = TAudioMath::MsToSamples(kSampleRate, 0.8f);
  const int seek_width = TAudioMath::MsToSamples(kSampleRate, 12.0f);
  const int max_seek = kFft / 2;
  const double* start = x;
  for (int i = 0; i < std::min(remaining, max_seek) - 1; ++i)
    if (x[i + 1] > x[i]) { start = x + i; remaining -= i; break; }
  const int seek_off = std::min(remaining, min_period);
  const double* end = start + seek_off;
  for (int i = 0; i < std::min(remaining - seek_off, max_seek) - 1; ++i)
    if (start[seek_off + i + 1] > start[seek_off + i]) { end = start + seek_off + i; break; }
  const int period = (int)(end - start);
  if (!remaining || period >= remaining) return 0.0;
  const int width = std::min(remaining, seek_width);
  std::vector<double> r(width, 0.0);
  TAutocorrelation::Calc(start, width, r.data(), width);
  double best = 0.0;
  for (int i = period / 2; i < width; ++i) best = std::max(best, r[i]);
  return best;
}

static int CmdNeighbours(const char* in, const char* outp, bool cap) {
  TRef R;
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int64_t nb = 0; if (fread(&nb, 8, 1, fi) != 1) return 1;
  std::vector<double> recs;
  int64_t total = 0;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t ns = 0; if (fread(&ns, 8, 1, fi) != 1) return 1;
    std::vector<double> x((size_t)ns);
    if (ns && fread(x.data(), 8, (size_t)ns, fi) != (size_t)ns) return 1;
    int64_t len = ns;
    if (cap) len = std::min<int64_t>(len, TAudioMath::MsToSamples(kSampleRate, 1000 * 20));
    // SampleAnalyser.cpp:798-809
    aubio_pitch_t* pitch = new_aubio_pitch("yinfast", kFft, kHop, kSampleRate);
    aubio_pitch_set_tolerance(pitch, 0.75);
    aubio_pitch_set_silence(pitch, -48.0);
    aubio_pitch_set_unit(pitch, "freq");
    aubio_spectral_whitening_t* white = new_aubio_spectral_whitening(kFft, kHop, kSampleRate);
    aubio_spectral_whitening_set_relax_time(white, 22);
    std::vector<double> mag(kFft, 0.0), wh(kFft, 0.0), peak(kFft, 0.0), harm(kFft, 0.0);
    std::vector<int> pbin(kFft); std::vector<double> pval(kFft);
    for (int64_t n = 0; (n + kFft - 1) < len; n += kHop) {
      recs.resize((size_t)(total + 1) * kNeigh);
      double* out = recs.data() + (size_t)total * kNeigh;
      R.Stft(x.data() + n, mag.data());
      fvec_t hop; hop.length = kHop; hop.data = x.data() + n;
      fvec_t frame; frame.length = kFft; frame.data = x.data() + n;
      // whitened spectrum (849-858)
      wh = mag;
      cvec_t grain; grain.length = kFft / 2; grain.norm = wh.data(); grain.phas = NULL;
      aubio_spectral_whitening_do(white, &grain);
      memcpy(out + nWhite, wh.data(), 1024 * sizeof(double));
      // peak spectrum (95-123, 861-862)
      const double thr = 0.25 * TStatistics::Max(wh.data(), kFft / 2);
      const int npk = PeakScan(wh.data(), kFft / 2, thr, pbin.data(), pval.data());
      peak = wh;
      for (int i = 0; i < kFft / 2; ++i) peak[i] = 0.0;
      for (int i = 0; i < npk; ++i) peak[pbin[i]] = pval[i];
      // silence (865-868)
      const bool silent = aubio_silence_detection(&hop, -48.0) == 1;
      out[nSilence] = silent ? 1.0 : 0.0;
      // envelope (1787-1804)
      {
        TEnvelopeDetector det(TEnvelopeDetector::kFast, 8.0, kSampleRate);
        double env = 0.0, top = 0.0;
        for (int i = 0; i < kHop; ++i) { det.Run(TMathT<double>::Abs(hop.data[i]), env); top = MMax(top, env); }
        out[nEnvelope] = top;
      }
      // F0 (876-917)
      double f0 = 0.0, conf = 0.0, safe = 0.0;
      {
        fvec_t po; po.length = 1; po.data = &f0;
        aubio_pitch_do(pitch, &frame, &po);
        conf = MClip(aubio_pitch_get_confidence(pitch) / 0.25, 0.0, 1.0);
        if (f0 > 0.0 && conf > 0.2) safe = f0;
        else if (!silent) {
          const double cb = TStatistics::Centroid(mag.data(), kFft / 2);
          safe = (double)kSampleRate / (double)kFft * MMax(cb, 0.0);
        }
      }
      out[nF0] = f0; out[nF0Conf] = conf; out[nF0FailSafe] = safe;
      // harmonic spectrum (920-927)
      { double a[4] = { 0 }; a[0] = safe; a[1] = 0.5; xtract_harmonic_spectrum(peak.data(), kFft, a, harm.data()); }
      // autocorrelation (943-944)
      out[nAutoCorr] = AutoCorrelation(x.data() + n, (int)(ns - n));
      // complexity (1937-1947), inharmonicity (1951-1971), tristimulus (1975-2003)
      { double c = 0.0; xtract_nonzero_count(peak.data() + R.mFirstBin, R.mBinCount, NULL, &c); out[nComplexity] = NanToZero(c); }
      out[nInharm] = out[nTri1] = out[nTri2] = out[nTri3] = 0.0;
      if (safe > 0.0 && conf > 0.0) {
        double v = 0.0; xtract_spectral_inharmonicity(peak.data(), kFft, &safe, &v); out[nInharm] = NanToZero(v);
        xtract_tristimulus_1(harm.data(), kFft, &safe, &out[nTri1]);
        xtract_tristimulus_2(harm.data(), kFft, &safe, &out[nTri2]);
        xtract_tristimulus_3(harm.data(), kFft, &safe, &out[nTri3]);
      }
      ++total;
    }
    del_aubio_pitch(pitch);
    del_aubio_spectral_whitening(white);
  }
  fclose(fi);
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  fwrite(&total, 8, 1, fo);
  fwrite(recs.data(), 8, recs.size(), fo);
  fclose(fo);
  return 0;
}

// ---- LoadSample front end, SampleAnalyser.cpp:484-718 ----
// The member body (decoder plumbing, TArray/TList buffers, logging) does not link here; its flow is restated
// on already-decoded interleaved PCM, with the reference's own TSampleConverter conversions
// (SampleConverter.h:446-449, 474-486, 529-533), TMathT<float>::GetMinMax, TAudioMath::DbToLin and constants.
// in.bin : int32 format (0 int16, 1 packed int24, 2 float32, 3 int32, 4 float64), int32 channels, int64 frames, raw samples
// out.bin: float peak, float rms, int32 data_offset, int32 lead, int32 trail, int32 pad, int64 n, double[n]
static int CmdLoad(const char* in, const char* outp) {
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int32_t format = 0, channels = 0; int64_t frames = 0;
  if (fread(&format, 4, 1, fi) != 1 || fread(&channels, 4, 1, fi) != 1 || fread(&frames, 8, 1, fi) != 1) return 1;
  const size_t bps = format == 0 ? 2 : (format == 1 ? 3 : (format == 4 ? 8 : (format == 5 ? 1 : 4)));   // 5: 8-bit unsigned
  std::vector<unsigned char> raw((size_t)frames * channels * bps);
  if (!raw.empty() && fread(raw.data(), 1, raw.size(), fi) != raw.size()) return 1;
  fclose(fi);
  // decoders hand "16-bit floats" per channel (WaveFile.cpp:384-410)
  std::vector<std::vector<float> > chan((size_t)channels, std::vector<float>((size_t)frames));
  for (int64_t n = 0; n < frames; ++n)
    for (int c = 0; c < channels; ++c) {
      const unsigned char* p = raw.data() + ((size_t)n * channels + c) * bps;
      float v;
      if (format == 0) { TInt16 s; memcpy(&s, p, 2); v = TSampleConverter::S16BitSignedTo16BitFloat(s); }
      else if (format == 1) { TSampleConverter::T24Pack t; t.mFirst = (TInt8)p[0]; t.mSecond = (TInt8)p[1]; t.mThird = (TInt8)p[2];
                              v = TSampleConverter::S24BitTo16BitFloat(t); }
      else if (format == 3) { TInt32 s; memcpy(&s, p, 4); v = TSampleConverter::S32BitSignedTo16BitFloat(s); }
      else if (format == 4) { double d; memcpy(&d, p, 8); v = TSampleConverter::S0To1FloatTo16BitFloat(d); }
      else if (format == 5) { v = TSampleConverter::S8BitUnsignedTo16BitFloat((TUInt8)p[0]); }
      else { float f; memcpy(&f, p, 4); v = TSampleConverter::S0To1FloatTo16BitFloat(f); }
      chan[(size_t)c][(size_t)n] = v;
    }
  // mono mix-down (SampleAnalyser.cpp:534-556)
  static const float sScaleFactor = M16BitSampleRange / 2.0f;
  int NumberOfSampleChannels = channels;
  const int NumberOfSampleFrames = (int)frames;
  if (NumberOfSampleChannels > 1) {
    float* pDestMonoBuffer = chan[0].data();
    const float MixDownScaling = 1.0f / (float)NumberOfSampleChannels;
    for (int n = 0; n < NumberOfSampleFrames; ++n) {
      for (int c = 1; c < NumberOfSampleChannels; ++c) pDestMonoBuffer[n] += chan[(size_t)c][(size_t)n];
      pDestMonoBuffer[n] *= MixDownScaling;
    }
    NumberOfSampleChannels = 1;
  }
  std::vector<float>& Buffer = chan[0];
  // rms, peak, amplification (SampleAnalyser.cpp:610-640)
  double RmsValue = 0.0;
  for (int n = 0; n < NumberOfSampleFrames; ++n) RmsValue += TMathT<double>::Square(Buffer[(size_t)n] / sScaleFactor);
  const float Rms = (float)MMin(1.0, ::sqrt(RmsValue / (NumberOfSampleChannels * NumberOfSampleFrames)));
  float Min = Buffer[0], Max = Buffer[0];
  TMathT<float>::GetMinMax(Min, Max, 0, NumberOfSampleFrames, Buffer.data());
  const double MaxAmplitude = (double)MMax(TMathT<float>::Abs(Min), TMathT<float>::Abs(Max));
  const float Peak = (float)MMin(1.0, MaxAmplitude / sScaleFactor);
  const double Amplification = (MaxAmplitude > MEpsilon) ? sScaleFactor / MaxAmplitude : 1.0;
  // silence trim (SampleAnalyser.cpp:646-670)
  static const double sSilenceFloor = sScaleFactor * TAudioMath::DbToLin(-48.0);
  int Lead = 0;
  for (int f = 0; f < NumberOfSampleFrames; ++f, ++Lead)
    if (TMathT<double>::Abs(Amplification * Buffer[(size_t)f]) > sSilenceFloor) break;
  int Trail = 0;
  for (int f = NumberOfSampleFrames - 1; f > Lead; --f, ++Trail)
    if (TMathT<double>::Abs(Amplification * Buffer[(size_t)f]) > sSilenceFloor) break;
  // padding (SampleAnalyser.cpp:679-718)
  const int Audible = NumberOfSampleFrames - Lead - Trail;
  int EndFrameOffset = 0;
  if ((Audible % kFft) < kFft / 2) EndFrameOffset += kFft / 2;
  int StartFrameOffset = 0;
  if (Audible + EndFrameOffset < kFft) StartFrameOffset = kFft - Audible - EndFrameOffset;
  std::vector<double> data((size_t)(Audible + StartFrameOffset + EndFrameOffset), 0.0);
  const double FinalScaling = (double)Amplification / sScaleFactor;
  for (int n = 0; n < Audible; ++n) data[(size_t)(n + StartFrameOffset)] = Buffer[(size_t)(n + Lead)] * FinalScaling;
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  const int32_t off = -Lead + StartFrameOffset, lead = Lead, trail = Trail, pad = 0; const int64_t n = (int64_t)data.size();
  fwrite(&Peak, 4, 1, fo); fwrite(&Rms, 4, 1, fo); fwrite(&off, 4, 1, fo); fwrite(&lead, 4, 1, fo); fwrite(&trail, 4, 1, fo);
  fwrite(&pad, 4, 1, fo); fwrite(&n, 8, 1, fo); fwrite(data.data(), 8, data.size(), fo);
  fclose(fo);
  return 0;
}

// The sample-rate conversion of LoadSample, SampleAnalyser.cpp:563-607: the reference's libresample driven by the call
// sequence of those lines (restated: LoadSample is a member of TSampleAnalyser, which does not link here).
// in:  int32 file_rate, int32 analyser_rate, int64 n, float[n]     out: int64 NewSize, int64 written, int64 used, float[NewSize]
static int CmdResample(const char* in, const char* outp) {
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int32_t file_rate = 0, rate = 0; int64_t n = 0;
  if (fread(&file_rate, 4, 1, fi) != 1 || fread(&rate, 4, 1, fi) != 1 || fread(&n, 8, 1, fi) != 1) return 1;
  std::vector<float> src((size_t)n);
  if (n > 0 && fread(src.data(), 4, (size_t)n, fi) != (size_t)n) return 1;
  fclose(fi);
  const double Speed = (double)file_rate / (double)rate;
  const unsigned int OldSizeInSamples = (unsigned int)n;
  const unsigned int NewSizeInSamples = (unsigned int)MMax(1, TMath::d2iRound((int)n / Speed));
  std::vector<float> dst((size_t)NewSizeInSamples, 0.0f);   // the reference's buffer is uninitialised (TArray::SetSize)
  const int HighQuality = 1;
  void* pResampler = ::resample_open(HighQuality, 1.0 / Speed, 1.0 / Speed);
  const int LastFlag = 1;
  int SrcSamplesUsed = 0;
  const int DestSamplesWritten = ::resample_process(pResampler, 1.0 / Speed, src.data(), OldSizeInSamples, LastFlag,
                                                    &SrcSamplesUsed, dst.data(), NewSizeInSamples);
  ::resample_close(pResampler);
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  const int64_t a = NewSizeInSamples, b = DestSamplesWritten, c = SrcSamplesUsed;
  fwrite(&a, 8, 1, fo); fwrite(&b, 8, 1, fo); fwrite(&c, 8, 1, fo); fwrite(dst.data(), 4, dst.size(), fo);
  fclose(fo);
  return 0;
}

// CalcEffectiveLength (SampleAnalyser.cpp:1715-1755): private member, flow restated around the reference's
// TAudioMath::DbToLin / SamplesToMs
static int CmdEffectiveLength(const char* in, const char* outp) {
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int64_t nb = 0; if (fread(&nb, 8, 1, fi) != 1) return 1;
  std::vector<double> res;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t ns = 0; if (fread(&ns, 8, 1, fi) != 1) return 1;
    std::vector<double> x((size_t)ns);
    if (ns && fread(x.data(), 8, (size_t)ns, fi) != (size_t)ns) return 1;
    const double Floors[3] = { TAudioMath::DbToLin(-48.0), TAudioMath::DbToLin(-24.0), TAudioMath::DbToLin(-12.0) };
    const int NumberOfSamples = (int)ns;
    for (int s = 0; s < 3; ++s) {
      int Lead = 0;
      for (int f = 0; f < NumberOfSamples; ++f, ++Lead)
        if (TMathT<double>::Abs(x[(size_t)f]) > Floors[s]) break;
      int Trail = 0;
      for (int f = NumberOfSamples - 1; f > Lead; --f, ++Trail)
        if (TMathT<double>::Abs(x[(size_t)f]) > Floors[s]) break;
      res.push_back(TAudioMath::SamplesToMs(kSampleRate, NumberOfSamples - Lead - Trail) / 1000.0);
    }
  }
  fclose(fi);
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  fwrite(res.data(), 8, res.size(), fo);
  fclose(fo);
  return 0;
}

// SToMsgpack (SqliteSampleDescriptorPool.cpp:596-713) for the low-level descriptors (kAllowBinaryStorage, doubles):
// the reference's own msgpack packer on a TList<double> (width 0) or a TList<TStaticArray<double, W>>.
// in.bin: int32 width, int32 pad, int64 rows, double[rows * max(width, 1)]; out.bin: the BLOB
static int CmdMsgpack(const char* in, const char* outp) {
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int32_t width = 0, pad = 0; int64_t rows = 0;
  if (fread(&width, 4, 1, fi) != 1 || fread(&pad, 4, 1, fi) != 1 || fread(&rows, 8, 1, fi) != 1) return 1;
  std::vector<double> v((size_t)rows * (size_t)std::max(width, 1));
  if (!v.empty() && fread(v.data(), 8, v.size(), fi) != v.size()) return 1;
  fclose(fi);
  msgpack::sbuffer Buffer;
  msgpack::packer<msgpack::sbuffer> Packer(&Buffer);
  Packer.pack_array((uint32_t)rows);
  if (width == 0) {
    for (int64_t i = 0; i < rows; ++i) Packer.pack(v[(size_t)i]);
  } else {
    for (int64_t i = 0; i < rows; ++i) {
      Packer.pack_array((uint32_t)width);
      for (int j = 0; j < width; ++j) Packer.pack(v[(size_t)i * width + j]);
    }
  }
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  fwrite(Buffer.data(), 1, Buffer.size(), fo);
  fclose(fo);
  return 0;
}

// C2 subset timing: window -> FFT -> magnitude -> xtract_mfcc on uniform noise.
static int CmdTime(int64_t nframes, unsigned seed) {
  TRef R;
  std::mt19937 gen(seed);
  std::uniform_real_distribution<float> U(-1.0f, 1.0f);
  std::vector<double> x((size_t)(nframes - 1) * kHop + kFft);
  for (auto& v : x) v = (double)U(gen);
  std::vector<double> mag(kFft), mf(kNumCep);
  double acc = 0.0;
  auto t0 = std::chrono::steady_clock::now();
  for (int64_t f = 0; f < nframes; ++f) {
    R.Stft(x.data() + f * kHop, mag.data());
    xtract_mfcc(mag.data(), kFft / 2, &R.mMel, mf.data());
    acc += mf[0];
  }
  auto t1 = std::chrono::steady_clock::now();
  const double s = std::chrono::duration<double>(t1 - t0).count();
  printf("{\"frames\": %lld, \"seconds\": %.6f, \"frames_per_s\": %.3f, \"checksum\": %.9g}\n",
         (long long)nframes, s, nframes / s, acc);
  return 0;
}

// TOnsetFftProcessor::LoadFrame (OnsetDetector.cpp:116-160) for every 512/128 frame of each buffer (the loop of
// SampleAnalyser.cpp:991-998, no duration cap): the class itself does not link here (TArray -> TMemory -> TString),
// so its flow is restated around the reference's own TFftWindow::SFillBuffer, ooura_cdft (as TFftTransformComplex's
// generic branch calls it, Fourier.cpp:243-262, kNoDiv), TAudioMath::Magnitude and TAudioMath::Phase.
// out.bin: int64 frames; per frame float[2 + 255 + 255] = mDC, mNyquist, mBin[].mMagn, mBin[].mPhase
static int CmdOnsetFft(const char* in, const char* outp) {
  const int Fft = 512, Hop = 128, Bins = Fft / 2 - 1;
  std::vector<double> Window(Fft), Re(Fft), Im(Fft), Inter(2 * Fft), W(Fft), Magnitude(Bins), Phase(Bins);
  std::vector<int> Ip(Fft);
  TFftWindow::SFillBuffer(TFftWindow::kHanning, Window.data(), Fft);
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int64_t nb = 0; if (fread(&nb, 8, 1, fi) != 1) return 1;
  std::vector<float> res;
  int64_t frames = 0;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t ns = 0; if (fread(&ns, 8, 1, fi) != 1) return 1;
    std::vector<double> x((size_t)ns);
    if (ns && fread(x.data(), 8, (size_t)ns, fi) != (size_t)ns) return 1;
    for (int64_t n = 0; (n + Fft - 1) < ns; n += Hop, ++frames) {
      memcpy(Re.data(), x.data() + n, Fft * sizeof(double));     // CopyBuffer / ClearBuffer are TMemory::Copy / Zero
      TAudioMath::MultiplyBuffers(Window.data(), Re.data(), Re.data(), Fft);
      memset(Im.data(), 0, Fft * sizeof(double));
      for (int i = 0; i < Fft; ++i) { Inter[2*i] = Re[i]; Inter[2*i+1] = Im[i]; }
      Ip[0] = 0;
      ooura_cdft(2 * Fft, 1, Inter.data(), Ip.data(), W.data());
      for (int i = 0; i < Fft; ++i) { Re[i] = Inter[2*i]; Im[i] = Inter[2*i+1]; }
      TAudioMath::Magnitude(Re.data(), Im.data(), Magnitude.data(), Bins);
      TAudioMath::Phase(Re.data(), Im.data(), Phase.data(), Bins);
      res.push_back((float)Re[0]);
      res.push_back((float)Im[0]);
      for (int i = 0; i < Bins; ++i) res.push_back((float)Magnitude[i]);
      for (int i = 0; i < Bins; ++i) res.push_back((float)Phase[i]);
    }
  }
  fclose(fi);
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  fwrite(&frames, 8, 1, fo);
  fwrite(res.data(), 4, res.size(), fo);
  fclose(fo);
  return 0;
}

// the aubio part of TRhythmTracker::CalculateTempo (RhythmTracker.cpp:176-197): a fresh tracker of the length of the
// (sharpened) onset series, one aubio_beattracking_do, bpm and confidence.  Each buffer of in.bin is one onset series.
// out.bin: per buffer double bpm, double confidence
static int CmdBeatTrack(const char* in, const char* outp) {
  FILE* fi = fopen(in, "rb"); if (!fi) return 1;
  int64_t nb = 0; if (fread(&nb, 8, 1, fi) != 1) return 1;
  std::vector<double> res;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t ns = 0; if (fread(&ns, 8, 1, fi) != 1) return 1;
    std::vector<double> Onsets((size_t)ns);
    if (ns && fread(Onsets.data(), 8, (size_t)ns, fi) != (size_t)ns) return 1;
    aubio_beattracking_t* pBeatTracker = ::new_aubio_beattracking((uint_t)ns, 128, kSampleRate);
    fvec_t BeatTrackInputVector;
    BeatTrackInputVector.length = (uint_t)ns;
    BeatTrackInputVector.data = Onsets.data();
    fvec_t* pBeatTrackOutputVector = new_fvec((uint_t)ns);
    ::aubio_beattracking_do(pBeatTracker, &BeatTrackInputVector, pBeatTrackOutputVector);
    res.push_back(::aubio_beattracking_get_bpm(pBeatTracker));
    res.push_back(::aubio_beattracking_get_confidence(pBeatTracker));
    ::del_fvec(pBeatTrackOutputVector);
    ::del_aubio_beattracking(pBeatTracker);
  }
  fclose(fi);
  FILE* fo = fopen(outp, "wb"); if (!fo) return 1;
  fwrite(res.data(), 8, res.size(), fo);
  fclose(fo);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "tables")) return CmdTables(argv[2]);
  if (argc >= 4 && !strcmp(argv[1], "frames")) return CmdFrames(argv[2], argv[3], argc >= 5 && atoi(argv[4]) != 0);
  if (argc >= 4 && !strcmp(argv[1], "neighbours")) return CmdNeighbours(argv[2], argv[3], argc >= 5 && atoi(argv[4]) != 0);
  if (argc >= 2 && !strcmp(argv[1], "peakstest")) return CmdPeaksTest();
  if (argc >= 4 && !strcmp(argv[1], "load")) return CmdLoad(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "resample")) return CmdResample(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "efflen")) return CmdEffectiveLength(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "msgpack")) return CmdMsgpack(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "onsetfft")) return CmdOnsetFft(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "beattrack")) return CmdBeatTrack(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "time")) return CmdTime(atoll(argv[2]), (unsigned)atoi(argv[3]));
  fprintf(stderr, "usage: ref_driver tables|frames|neighbours|load|efflen|msgpack|onsetfft|beattrack|peakstest|time ...\n");
  return 2;
}
