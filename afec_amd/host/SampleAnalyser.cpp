// afec_amd/host/SampleAnalyser.cpp -- see SampleAnalyser.h.
#include "SampleAnalyser.h"

#include <algorithm>
#include <chrono>
#include <cmath>

#include "../../include/afx.h"

namespace afec {

namespace {

[[noreturn]] void Throw(const char* What, int Status) {
  throw TReadableException(std::string(What) + ": " + afx_status_str(Status) + " (" + afx_last_error() + ")");
}

}  // namespace

TSampleAnalyser::TSampleAnalyser(int SampleRate, int FftFrameSize, int HopFrameSize, int Device, int FrameKernel)
    : mpPlan(nullptr), mSampleRate(SampleRate), mFftFrameSize(FftFrameSize), mHopFrameSize(HopFrameSize) {
  afx_plan_desc Desc = {SampleRate, FftFrameSize, HopFrameSize, Device, AFX_PRECISION_F64,
                        /* MAnalyzationDurationMaxInMs, SampleAnalyser.cpp:37 */ 1000 * 20, FrameKernel};
  const int Status = afx_plan_create(&Desc, &mpPlan);
  if (Status != AFX_OK) Throw("GPU feature extraction unavailable", Status);
}

TSampleAnalyser::~TSampleAnalyser() { afx_plan_destroy(mpPlan); }

void TSampleAnalyser::SetSleepingWaits(bool Sleeping) { afx_plan_set_blocking_wait(mpPlan, Sleeping ? 1 : 0); }

int64_t TSampleAnalyser::NumberOfFrames(int64_t NumberOfSamples) const { return afx_num_frames(mpPlan, NumberOfSamples); }

namespace {
template <int W>
void Fill(TFramedVectorData<W>& Dst, const double* pSrc, int64_t Frames) {
  Dst.mValues.resize((size_t)Frames);
  for (int64_t f = 0; f < Frames; ++f)
    for (int b = 0; b < W; ++b) Dst.mValues[(size_t)f][b] = pSrc[f * W + b];
}
void Fill(TFramedScalarData& Dst, const double* pSrc, int64_t Frames) { Dst.mValues.assign(pSrc, pSrc + Frames); }
}  // namespace

namespace {
// the 13 values of TStatistics::Calc in the order of AFX_S_* -> the members of the framed data
void FillStatistics(TFramedScalarData& Dst, const double* pS) {
  Dst.mMin = pS[AFX_S_MIN]; Dst.mMax = pS[AFX_S_MAX]; Dst.mMedian = pS[AFX_S_MEDIAN]; Dst.mMean = pS[AFX_S_MEAN];
  Dst.mGeometricMean = pS[AFX_S_GMEAN]; Dst.mVariance = pS[AFX_S_VARIANCE]; Dst.mCentroid = pS[AFX_S_CENTROID];
  Dst.mSpread = pS[AFX_S_SPREAD]; Dst.mSkewness = pS[AFX_S_SKEWNESS]; Dst.mKurtosis = pS[AFX_S_KURTOSIS];
  Dst.mFlatness = pS[AFX_S_FLATNESS]; Dst.mDMean = pS[AFX_S_DMEAN]; Dst.mDVariance = pS[AFX_S_DVARIANCE];
}
template <int W>
void FillStatistics(TFramedVectorData<W>& Dst, const double* pS) {
  for (int b = 0; b < W; ++b) {
    const double* p = pS + (size_t)b * AFX_NUM_STATISTICS;
    Dst.mMin[b] = p[AFX_S_MIN]; Dst.mMax[b] = p[AFX_S_MAX]; Dst.mMedian[b] = p[AFX_S_MEDIAN]; Dst.mMean[b] = p[AFX_S_MEAN];
    Dst.mGeometricMean[b] = p[AFX_S_GMEAN]; Dst.mVariance[b] = p[AFX_S_VARIANCE]; Dst.mCentroid[b] = p[AFX_S_CENTROID];
    Dst.mSpread[b] = p[AFX_S_SPREAD]; Dst.mSkewness[b] = p[AFX_S_SKEWNESS]; Dst.mKurtosis[b] = p[AFX_S_KURTOSIS];
    Dst.mFlatness[b] = p[AFX_S_FLATNESS]; Dst.mDMean[b] = p[AFX_S_DMEAN]; Dst.mDVariance[b] = p[AFX_S_DVARIANCE];
  }
}

// one series of the ABI: per-frame values and per-buffer statistics, and where they go in TSampleDescriptors
struct TScalarSeries {
  double* afx_out::*mpOut;
  double* afx_stats_out::*mpStat;
  TFramedScalarData TSampleDescriptors::*mpDst;
};
template <int W>
struct TVectorSeries {
  double* afx_out::*mpOut;
  double* afx_stats_out::*mpStat;
  TFramedVectorData<W> TSampleDescriptors::*mpDst;
};
#define AFEC_SERIES(abi, member) {&afx_out::abi, &afx_stats_out::abi, &TSampleDescriptors::member}
const TScalarSeries kScalarSeries[] = {
    AFEC_SERIES(amplitude_silence, mAmplitudeSilence), AFEC_SERIES(amplitude_peak, mAmplitudePeak),
    AFEC_SERIES(amplitude_rms, mAmplitudeRms), AFEC_SERIES(amplitude_envelope, mAmplitudeEnvelope),
    AFEC_SERIES(spectral_rms, mSpectralRms), AFEC_SERIES(spectral_centroid, mSpectralCentroid),
    AFEC_SERIES(spectral_rolloff, mSpectralRolloff), AFEC_SERIES(spectral_spread, mSpectralSpread),
    AFEC_SERIES(spectral_skewness, mSpectralSkewness), AFEC_SERIES(spectral_kurtosis, mSpectralKurtosis),
    AFEC_SERIES(spectral_flatness, mSpectralFlatness), AFEC_SERIES(spectral_inharmonicity, mSpectralInharmonicity),
    AFEC_SERIES(spectral_complexity, mSpectralComplexity), AFEC_SERIES(spectral_contrast, mSpectralContrast),
    AFEC_SERIES(spectral_flux, mSpectralFlux), AFEC_SERIES(f0, mF0), AFEC_SERIES(f0_confidence, mF0Confidence),
    AFEC_SERIES(failsafe_f0, mFailSafeF0), AFEC_SERIES(tristimulus1, mTristimulus1),
    AFEC_SERIES(tristimulus2, mTristimulus2), AFEC_SERIES(tristimulus3, mTristimulus3),
    AFEC_SERIES(auto_correlation, mAutoCorrelation)};
const TVectorSeries<14> kSubBandSeries[] = {
    AFEC_SERIES(sub_rms, mSpectralRmsBands), AFEC_SERIES(sub_flatness, mSpectralFlatnessBands),
    AFEC_SERIES(sub_flux, mSpectralFluxBands), AFEC_SERIES(sub_complexity, mSpectralComplexityBands),
    AFEC_SERIES(sub_contrast, mSpectralContrastBands), AFEC_SERIES(mfcc, mCepstrumBands)};
const TVectorSeries<28> kBandSeries[] = {AFEC_SERIES(spectrum_bands, mSpectrumBands)};
#undef AFEC_SERIES

// the rhythm tracker's results of file i (afx_batch_fetch_rhythm layout) -> TSampleDescriptors
void FillRhythm(TSampleDescriptors& R, int i, const int64_t* pOffset, const double* pOnsets, const double* pScalars,
                const double* pStatistics) {
  const int64_t t0 = pOffset[i], nt = pOffset[i + 1] - t0;
  R.mRhythmComplexOnsets.mValues.resize((size_t)nt);
  R.mRhythmPercussiveOnsets.mValues.resize((size_t)nt);
  for (int64_t t = 0; t < nt; ++t) {
    R.mRhythmComplexOnsets.mValues[(size_t)t] = pOnsets[(t0 + t) * 2];
    R.mRhythmPercussiveOnsets.mValues[(size_t)t] = pOnsets[(t0 + t) * 2 + 1];
  }
  FillStatistics(R.mRhythmComplexOnsets, pStatistics + (size_t)i * 2 * AFX_NUM_STATISTICS);
  FillStatistics(R.mRhythmPercussiveOnsets, pStatistics + ((size_t)i * 2 + 1) * AFX_NUM_STATISTICS);
  const double* s = pScalars + (size_t)i * AFX_NUM_RHYTHM_SCALARS;
  R.mRhythmComplexOnsetCount = s[AFX_R_COMPLEX_ONSET_COUNT];
  R.mRhythmComplexTempo = s[AFX_R_COMPLEX_TEMPO];
  R.mRhythmComplexTempoConfidence = s[AFX_R_COMPLEX_TEMPO_CONFIDENCE];
  R.mRhythmComplexOnsetFrequencyMean = s[AFX_R_COMPLEX_ONSET_FREQUENCY_MEAN];
  R.mRhythmComplexOnsetStrength = s[AFX_R_COMPLEX_ONSET_STRENGTH];
  R.mRhythmComplexOnsetContrast = s[AFX_R_COMPLEX_ONSET_CONTRAST];
  R.mRhythmPercussiveOnsetCount = s[AFX_R_PERCUSSIVE_ONSET_COUNT];
  R.mRhythmPercussiveTempo = s[AFX_R_PERCUSSIVE_TEMPO];
  R.mRhythmPercussiveTempoConfidence = s[AFX_R_PERCUSSIVE_TEMPO_CONFIDENCE];
  R.mRhythmPercussiveOnsetFrequencyMean = s[AFX_R_PERCUSSIVE_ONSET_FREQUENCY_MEAN];
  R.mRhythmPercussiveOnsetStrength = s[AFX_R_PERCUSSIVE_ONSET_STRENGTH];
  R.mRhythmPercussiveOnsetContrast = s[AFX_R_PERCUSSIVE_ONSET_CONTRAST];
  R.mRhythmFinalTempo = s[AFX_R_FINAL_TEMPO];
  R.mRhythmFinalTempoConfidence = s[AFX_R_FINAL_TEMPO_CONFIDENCE];
}

// TSampleData::mOriginalSampleRate / mOriginalNumberOfSamples of files the caller resampled (SampleAnalyser.cpp:464-467)
int SetFileInfo(afx_batch* pBatch, const std::vector<TDecodedSample>& Files, const std::vector<afx_load_info>& Info, int PlanRate) {
  bool Any = false;
  for (const TDecodedSample& f : Files) Any = Any || f.mOriginalSampleRate > 0 || f.mOriginalNumberOfSamples > 0;
  if (!Any) return AFX_OK;
  std::vector<afx_file_info> FileInfo(Files.size());
  for (size_t i = 0; i < Files.size(); ++i) {
    FileInfo[i].original_sample_rate = Files[i].mOriginalSampleRate > 0 ? Files[i].mOriginalSampleRate : PlanRate;
    FileInfo[i].original_samples = Files[i].mOriginalNumberOfSamples > 0 ? Files[i].mOriginalNumberOfSamples : Files[i].mNumberOfSampleFrames;
    FileInfo[i].data_offset = Info[i].data_offset;
  }
  return afx_batch_set_file_info(pBatch, FileInfo.data());
}

struct TBatchGuard {
  afx_batch* mpBatch = nullptr;
  ~TBatchGuard() { afx_batch_destroy(mpBatch); }
};
}  // namespace

// The whole of AnalyzeLowLevelDescriptors' loop (SampleAnalyser.cpp:814-976) and of CalcStatistics
// (SampleAnalyser.cpp:1065, 2402-2412) runs on the GPU: every per-frame series and its 13 statistics come
// back from one resident batch.
namespace {
constexpr uint32_t kEverything = AFX_D_ALL_PER_FRAME | AFX_D_EFFECTIVE_LENGTH | AFX_D_RHYTHM | AFX_D_STATISTICS;
std::vector<TSampleDescriptors> Collect(TBatchGuard& Batch, int32_t n, std::vector<std::string>* pFailed);
}  // namespace

std::vector<TSampleDescriptors> TSampleAnalyser::AnalyzeLowLevelDescriptors(
    const std::vector<const std::vector<double>*>& Samples, std::vector<std::string>* pFailed) const {
  const int32_t n = (int32_t)Samples.size();
  std::vector<afx_buf> Buffers((size_t)n);
  for (int32_t i = 0; i < n; ++i) Buffers[i] = {Samples[i]->data(), AFX_PCM_F64, 0, (int64_t)Samples[i]->size()};
  TBatchGuard Batch;
  int Status = afx_batch_create(mpPlan, Buffers.data(), n, kEverything, &Batch.mpBatch);
  if (Status == AFX_OK) Status = afx_batch_run(Batch.mpBatch);
  if (Status != AFX_OK) Throw("GPU feature extraction failed", Status);
  return Collect(Batch, n, pFailed);
}

// LoadSample (SampleAnalyser.cpp:443-719, after the container decode) + AnalyzeLowLevelDescriptors +
// CalcStatistics for decoded files: conversion, mono mix-down, normalisation, silence trim and padding run
// on the GPU as well (afx_batch_create_from_raw)
std::vector<TSampleDescriptors> TSampleAnalyser::Analyze(const std::vector<TDecodedSample>& Files,
                                                         std::vector<TSampleDataInfo>* pInfo,
                                                         std::vector<std::string>* pFailed) const {
  const int32_t n = (int32_t)Files.size();
  std::vector<afx_raw> Raws((size_t)n);
  for (int32_t i = 0; i < n; ++i)
    Raws[i] = {Files[i].mpInterleavedSamples, Files[i].mFormat, Files[i].mNumberOfChannels, Files[i].mSampleRate, 0,
               Files[i].mNumberOfSampleFrames};
  std::vector<afx_load_info> Info((size_t)n);
  TBatchGuard Batch;
  int Status = afx_batch_create_from_raw(mpPlan, Raws.data(), n, kEverything, &Batch.mpBatch, Info.data());
  if (Status == AFX_OK) Status = SetFileInfo(Batch.mpBatch, Files, Info, mSampleRate);
  if (Status == AFX_OK) Status = afx_batch_run(Batch.mpBatch);
  if (Status != AFX_OK) Throw("GPU feature extraction failed", Status);
  if (pInfo) {
    pInfo->resize((size_t)n);
    for (int32_t i = 0; i < n; ++i)
      (*pInfo)[i] = {Info[i].peak_value, Info[i].rms_value, Info[i].data_offset, Info[i].n_samples};
  }
  return Collect(Batch, n, pFailed);
}

namespace {
std::vector<TSampleDescriptors> Collect(TBatchGuard& Batch, int32_t n, std::vector<std::string>* pFailed) {
  int Status = AFX_OK;
  const size_t F = (size_t)afx_batch_total_frames(Batch.mpBatch);

  // host arrays for every series: [F][W] values and [n][W][13] statistics
  std::vector<std::vector<double>> Values, Stats;
  afx_out Out = {};
  afx_stats_out StatsOut = {};
  auto Bind = [&](double* afx_out::*pOut, double* afx_stats_out::*pStat, int W) {
    Values.emplace_back(F * (size_t)W);
    Stats.emplace_back((size_t)n * (size_t)W * AFX_NUM_STATISTICS);
    Out.*pOut = Values.back().data();
    StatsOut.*pStat = Stats.back().data();
  };
  for (const TScalarSeries& S : kScalarSeries) Bind(S.mpOut, S.mpStat, 1);
  for (const TVectorSeries<14>& S : kSubBandSeries) Bind(S.mpOut, S.mpStat, 14);
  for (const TVectorSeries<28>& S : kBandSeries) Bind(S.mpOut, S.mpStat, 28);
  std::vector<int64_t> Offset((size_t)n + 1);
  std::vector<int32_t> BufStatus((size_t)n), StatsStatus((size_t)n);
  std::vector<double> EffectiveLength((size_t)n * 3);
  Out.effective_length = EffectiveLength.data();
  Out.frame_offset = Offset.data();
  Out.buf_status = BufStatus.data();
  StatsOut.stats_status = StatsStatus.data();
  Status = afx_batch_fetch(Batch.mpBatch, &Out);
  if (Status == AFX_OK) Status = afx_batch_fetch_statistics(Batch.mpBatch, &StatsOut);
  std::vector<int64_t> RhythmOffset((size_t)n + 1);
  const size_t RhythmRows = (size_t)afx_batch_rhythm_frames(Batch.mpBatch, RhythmOffset.data());
  std::vector<double> Onsets(RhythmRows * 2), RhythmScalars((size_t)n * AFX_NUM_RHYTHM_SCALARS),
      RhythmStatistics((size_t)n * 2 * AFX_NUM_STATISTICS);
  if (Status == AFX_OK) Status = afx_batch_fetch_rhythm(Batch.mpBatch, Onsets.data(), RhythmScalars.data(), RhythmStatistics.data());
  if (Status != AFX_OK) Throw("GPU feature extraction failed", Status);

  std::vector<TSampleDescriptors> Results((size_t)n);
  if (pFailed) pFailed->assign((size_t)n, std::string());
  for (int32_t i = 0; i < n; ++i) {
    const int32_t Bad = (BufStatus[i] != AFX_OK) ? BufStatus[i] : StatsStatus[i];
    if (Bad != AFX_OK) {
      if (pFailed) (*pFailed)[i] = std::string("Sample failed to load: ") + afx_status_str(Bad);
      continue;
    }
    const int64_t f0 = Offset[i], nf = Offset[i + 1] - Offset[i];
    TSampleDescriptors& R = Results[i];
    R.mEffectiveLength48dB = EffectiveLength[(size_t)i * 3];
    R.mEffectiveLength24dB = EffectiveLength[(size_t)i * 3 + 1];
    R.mEffectiveLength12dB = EffectiveLength[(size_t)i * 3 + 2];
    for (const TScalarSeries& S : kScalarSeries) {
      Fill(R.*(S.mpDst), (Out.*(S.mpOut)) + f0, nf);
      FillStatistics(R.*(S.mpDst), (StatsOut.*(S.mpStat)) + (size_t)i * AFX_NUM_STATISTICS);
    }
    for (const TVectorSeries<14>& S : kSubBandSeries) {
      Fill(R.*(S.mpDst), (Out.*(S.mpOut)) + f0 * 14, nf);
      FillStatistics(R.*(S.mpDst), (StatsOut.*(S.mpStat)) + (size_t)i * 14 * AFX_NUM_STATISTICS);
    }
    for (const TVectorSeries<28>& S : kBandSeries) {
      Fill(R.*(S.mpDst), (Out.*(S.mpOut)) + f0 * 28, nf);
      FillStatistics(R.*(S.mpDst), (StatsOut.*(S.mpStat)) + (size_t)i * 28 * AFX_NUM_STATISTICS);
    }
    FillRhythm(R, i, RhythmOffset.data(), Onsets.data(), RhythmScalars.data(), RhythmStatistics.data());
  }
  return Results;
}
}  // namespace

namespace {
// the series of afx_batch_record_layout, in its order
double* afx_out::* const kRecordOrder[AFX_NUM_SERIES] = {
    &afx_out::mfcc, &afx_out::spectral_rms, &afx_out::spectral_centroid, &afx_out::spectral_spread, &afx_out::spectral_skewness,
    &afx_out::spectral_kurtosis, &afx_out::spectral_rolloff, &afx_out::spectral_flatness, &afx_out::spectral_flux,
    &afx_out::spectrum_bands, &afx_out::amplitude_peak, &afx_out::amplitude_rms, &afx_out::sub_rms, &afx_out::sub_flatness,
    &afx_out::sub_flux, &afx_out::sub_complexity, &afx_out::sub_contrast, &afx_out::spectral_contrast,
    &afx_out::amplitude_silence, &afx_out::amplitude_envelope, &afx_out::spectral_complexity, &afx_out::auto_correlation,
    &afx_out::f0, &afx_out::f0_confidence, &afx_out::failsafe_f0, &afx_out::spectral_inharmonicity, &afx_out::tristimulus1,
    &afx_out::tristimulus2, &afx_out::tristimulus3};
int SeriesIndex(double* afx_out::*pMember) {
  for (int i = 0; i < AFX_NUM_SERIES; ++i)
    if (kRecordOrder[i] == pMember) return i;
  return -1;
}
void FillStrided(TFramedScalarData& Dst, const double* pSrc, int64_t Frames, int Stride) {
  Dst.mValues.resize((size_t)Frames);
  for (int64_t f = 0; f < Frames; ++f) Dst.mValues[(size_t)f] = pSrc[f * Stride];
}
template <int W>
void FillStrided(TFramedVectorData<W>& Dst, const double* pSrc, int64_t Frames, int Stride) {
  Dst.mValues.resize((size_t)Frames);
  for (int64_t f = 0; f < Frames; ++f)
    for (int b = 0; b < W; ++b) Dst.mValues[(size_t)f][b] = pSrc[f * Stride + b];
}
}  // namespace

TSampleDescriptors TRecordBatch::Descriptors(int i) const {
  TSampleDescriptors R;
  const int64_t f0 = mFrameOffset[(size_t)i], nf = mFrameOffset[(size_t)i + 1] - f0;
  R.mEffectiveLength48dB = mEffectiveLength[(size_t)i * 3];
  R.mEffectiveLength24dB = mEffectiveLength[(size_t)i * 3 + 1];
  R.mEffectiveLength12dB = mEffectiveLength[(size_t)i * 3 + 2];
  const double* const pRows = mpRecords + f0 * mStride;
  const double* const pStats = mpStatistics + (size_t)i * mStride * AFX_NUM_STATISTICS;
  auto Each = [&](auto& Series) {
    for (const auto& S : Series) {
      const int k = SeriesIndex(S.mpOut);
      if (k < 0 || mOffsets[k] < 0) continue;
      FillStrided(R.*(S.mpDst), pRows + mOffsets[k], nf, mStride);
      FillStatistics(R.*(S.mpDst), pStats + (size_t)mOffsets[k] * AFX_NUM_STATISTICS);
    }
  };
  Each(kScalarSeries);
  Each(kSubBandSeries);
  Each(kBandSeries);
  if (!mRhythmOffset.empty() && mpRhythmScalars)
    FillRhythm(R, i, mRhythmOffset.data(), mpRhythmOnsets, mpRhythmScalars, mpRhythmStatistics);
  return R;
}

// onsets [rows][2] + scalars [n][14] + onset statistics [n][2][13]; LoadSample pads a file by at most one 2048-sample
// frame, so it has at most samples / 128 + 17 rows
// sample frames of a file once it is at `Rate` (NewSizeInSamples, SampleAnalyser.cpp:572-573)
int64_t TSampleAnalyser::ConvertedSampleFrames(const TDecodedSample& File, int Rate) {
  if (File.mSampleRate <= 0 || File.mSampleRate == Rate) return File.mNumberOfSampleFrames;
  return (int64_t)((double)File.mNumberOfSampleFrames / ((double)File.mSampleRate / (double)Rate) + 0.5) + 1;
}

bool TSampleAnalyser::DeviceUsable() const {
  // "out of memory" is an answer: the device is alive, this batch was too large for what is free (retried in halves)
  const int Status = afx_plan_probe_device(mpPlan);
  return Status == AFX_OK || Status == AFX_ERR_OUT_OF_MEMORY;
}

size_t TSampleAnalyser::RhythmDoubles(const std::vector<TDecodedSample>& Files) const {
  size_t Rows = 0;
  for (const TDecodedSample& f : Files) Rows += (size_t)(ConvertedSampleFrames(f, mSampleRate) / 128) + 17;
  return Rows * 2 + Files.size() * (AFX_NUM_RHYTHM_SCALARS + 2 * AFX_NUM_STATISTICS);
}

bool TSampleAnalyser::AnalyzeToRecords(const std::vector<TDecodedSample>& Files, double* pRecords, size_t RecordCapacity,
                                       double* pStatistics, double* pRhythm, size_t RhythmCapacity, TRecordBatch& Result) const {
  const int32_t n = (int32_t)Files.size();
  std::vector<afx_raw> Raws((size_t)n);
  for (int32_t i = 0; i < n; ++i)
    Raws[i] = {Files[i].mpInterleavedSamples, Files[i].mFormat, Files[i].mNumberOfChannels, Files[i].mSampleRate, 0,
               Files[i].mNumberOfSampleFrames};
  std::vector<afx_load_info> Info((size_t)n);
  TBatchGuard Batch;
  auto Now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = Now();
  int Status = afx_batch_create_from_raw(mpPlan, Raws.data(), n, kEverything, &Batch.mpBatch, Info.data());
  if (Status != AFX_OK) Throw("GPU feature extraction failed", Status);
  const double t1 = Now();
  Result = TRecordBatch();
  afx_batch_record_layout(Batch.mpBatch, &Result.mStride, Result.mOffsets, Result.mWidths);
  const size_t Frames = (size_t)afx_batch_total_frames(Batch.mpBatch);
  if (Result.mStride > kMaxStride) throw TReadableException("AnalyzeToRecords: record stride exceeds kMaxStride");   // larger buffers would not help
  if (Frames * (size_t)Result.mStride > RecordCapacity) return false;
  Result.mRhythmOffset.resize((size_t)n + 1);
  const size_t RhythmRows = (size_t)afx_batch_rhythm_frames(Batch.mpBatch, Result.mRhythmOffset.data());
  if (RhythmRows * 2 + (size_t)n * (AFX_NUM_RHYTHM_SCALARS + 2 * AFX_NUM_STATISTICS) > RhythmCapacity) return false;
  Status = SetFileInfo(Batch.mpBatch, Files, Info, mSampleRate);
  if (Status == AFX_OK) Status = afx_batch_run(Batch.mpBatch);
  const double t2 = Now();
  Result.mFrameOffset.resize((size_t)n + 1);
  Result.mStatus.resize((size_t)n);
  Result.mEffectiveLength.resize((size_t)n * 3);
  if (Status == AFX_OK)
    Status = afx_batch_fetch_records(Batch.mpBatch, pRecords, pStatistics, Result.mFrameOffset.data(), Result.mStatus.data(),
                                     Result.mEffectiveLength.data());
  double* const pOnsets = pRhythm;
  double* const pScalars = pOnsets + RhythmRows * 2;
  double* const pOnsetStatistics = pScalars + (size_t)n * AFX_NUM_RHYTHM_SCALARS;
  if (Status == AFX_OK) Status = afx_batch_fetch_rhythm(Batch.mpBatch, pOnsets, pScalars, pOnsetStatistics);
  Result.mpRhythmOnsets = pOnsets; Result.mpRhythmScalars = pScalars; Result.mpRhythmStatistics = pOnsetStatistics;
  if (Status != AFX_OK) Throw("GPU feature extraction failed", Status);
  Result.mSeconds[0] = t1 - t0; Result.mSeconds[1] = t2 - t1; Result.mSeconds[2] = Now() - t2;
  Result.mpRecords = pRecords;
  Result.mpStatistics = pStatistics;
  Result.mInfo.resize((size_t)n);
  for (int32_t i = 0; i < n; ++i) Result.mInfo[i] = {Info[i].peak_value, Info[i].rms_value, Info[i].data_offset, Info[i].n_samples};
  return true;
}

TSampleDescriptors TSampleAnalyser::AnalyzeLowLevelDescriptors(const std::vector<double>& SampleData,
                                                               bool WithMagnitudes) const {
  std::vector<std::string> Failed;
  std::vector<TSampleDescriptors> R = AnalyzeLowLevelDescriptors({&SampleData}, &Failed);
  if (!Failed[0].empty()) throw TReadableException(Failed[0]);
  if (WithMagnitudes) {
    const int64_t nf = NumberOfFrames((int64_t)SampleData.size());
    R[0].mMagnitudeSpectrum.resize((size_t)nf * 1024);
    afx_buf Buffer = {SampleData.data(), AFX_PCM_F64, 0, (int64_t)SampleData.size()};
    afx_out Out = {};
    Out.magnitude = R[0].mMagnitudeSpectrum.data();
    const int Status = afx_extract_batch(mpPlan, &Buffer, 1, AFX_D_MAGNITUDE, &Out);
    if (Status != AFX_OK) Throw("GPU feature extraction failed", Status);
  }
  return std::move(R[0]);
}

}  // namespace afec
