// afec_amd/host/SampleAnalyser.h -- C++ host side above the C-ABI (include/afx.h), mirroring the
// reference's interface for this path: TSampleAnalyser (Export/SampleAnalyser.h:33-63) and the
// low-level part of TSampleDescriptors (Export/SampleDescriptors.h:152-356, 395-465).  Same names,
// same argument meaning, errors as exceptions (TReadableException there, std::runtime_error here).
//
// Only what the GPU path produces is present: every low-level descriptor and its statistics, the rhythm tracker's
// (SampleAnalyser.cpp:985-1048) included.  This layer computes nothing itself: values and statistics are the GPU's,
// fetched through the C-ABI.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

struct afx_plan;

namespace afec {

class TReadableException : public std::runtime_error {
public:
  explicit TReadableException(const std::string& What) : std::runtime_error(What) {}
};

// TSampleDescriptors::TFramedScalarData (SampleDescriptors.h:152-230): values per frame plus the
// 13 statistics TStatistics::Calc fills
struct TFramedScalarData {
  std::vector<double> mValues;  // [frame]
  double mMin = 0, mMax = 0, mMedian = 0, mMean = 0, mGeometricMean = 0, mVariance = 0, mCentroid = 0,
         mSpread = 0, mSkewness = 0, mKurtosis = 0, mFlatness = 0, mDMean = 0, mDVariance = 0;
};

// TSampleDescriptors::TFramedVectorData<W> (SampleDescriptors.h:262-356): [frame][band] values plus
// per-band statistics
template <int W>
struct TFramedVectorData {
  std::vector<std::array<double, W>> mValues;  // [frame][band]
  std::array<double, W> mMin{}, mMax{}, mMedian{}, mMean{}, mGeometricMean{}, mVariance{}, mCentroid{},
      mSpread{}, mSkewness{}, mKurtosis{}, mFlatness{}, mDMean{}, mDVariance{};
};

// the low-level descriptors of TSampleDescriptors that lie on the GPU path
struct TSampleDescriptors {
  enum { kNumberOfSpectrumSubBands = 14, kNumberOfSpectrumBands = 28, kNumberOfCepstrumCoefficients = 14 };

  // seconds between the first and last sample above -48 / -24 / -12 dB (SampleAnalyser.cpp:1715-1755)
  double mEffectiveLength48dB = 0, mEffectiveLength24dB = 0, mEffectiveLength12dB = 0;
  TFramedScalarData mAmplitudeSilence, mAmplitudePeak, mAmplitudeRms, mAmplitudeEnvelope;
  TFramedScalarData mF0, mF0Confidence, mFailSafeF0, mAutoCorrelation;
  TFramedScalarData mSpectralComplexity, mSpectralInharmonicity, mTristimulus1, mTristimulus2, mTristimulus3;
  TFramedScalarData mSpectralRms, mSpectralCentroid, mSpectralRolloff, mSpectralSpread, mSpectralSkewness,
      mSpectralKurtosis, mSpectralFlatness, mSpectralContrast, mSpectralFlux;
  TFramedVectorData<kNumberOfSpectrumSubBands> mSpectralRmsBands, mSpectralFlatnessBands, mSpectralFluxBands,
      mSpectralComplexityBands, mSpectralContrastBands;
  TFramedVectorData<kNumberOfSpectrumBands> mSpectrumBands;
  TFramedVectorData<kNumberOfCepstrumCoefficients> mCepstrumBands;

  // rhythm tracker (SampleDescriptors.h:434-452; SampleAnalyser.cpp:1006-1048): one onset value per 512/128 frame
  TFramedScalarData mRhythmComplexOnsets, mRhythmPercussiveOnsets;
  double mRhythmComplexOnsetCount = 0, mRhythmComplexOnsetFrequencyMean = 0, mRhythmComplexOnsetStrength = 0,
         mRhythmComplexOnsetContrast = 0, mRhythmComplexTempo = 0, mRhythmComplexTempoConfidence = 0;
  double mRhythmPercussiveOnsetCount = 0, mRhythmPercussiveOnsetFrequencyMean = 0, mRhythmPercussiveOnsetStrength = 0,
         mRhythmPercussiveOnsetContrast = 0, mRhythmPercussiveTempo = 0, mRhythmPercussiveTempoConfidence = 0;
  double mRhythmFinalTempo = 0, mRhythmFinalTempoConfidence = 0;

  // magnitude spectra [frame][1024] for the CPU-resident neighbours (optional)
  std::vector<double> mMagnitudeSpectrum;
};

// one decoded file as the reference's decoders hand it to LoadSample: interleaved PCM (SampleAnalyser.cpp:484-530)
struct TDecodedSample {
  const void* mpInterleavedSamples;
  int mFormat;              // AFX_RAW_I16 / AFX_RAW_I24 / AFX_RAW_F32 of include/afx.h
  int mNumberOfChannels;    // 1..8
  int mSampleRate;          // 0 = the analyser's rate; other rates are converted on the GPU (SampleAnalyser.cpp:563-607)
  int64_t mNumberOfSampleFrames;
  // what the file looked like before the caller resampled it (TSampleData::mOriginalSampleRate / mOriginalNumberOfSamples,
  // SampleAnalyser.cpp:464-467; the final tempo's duration heuristics use them); 0 = as given above
  int mOriginalSampleRate = 0;
  int64_t mOriginalNumberOfSamples = 0;
};
// what LoadSample leaves in TSampleData besides the samples (Export/SampleAnalyser.h:75-100)
struct TSampleDataInfo {
  float mPeakValue, mRmsValue;
  int mDataOffset;
  int64_t mNumberOfSamples;   // size of the normalised, trimmed, padded mono buffer
};

// One GPU batch's results as they come off the device (afx_batch_fetch_records): per-frame records
// double[frames][stride], statistics double[files][stride][13].  The buffers are the caller's (page-locked memory
// keeps the transfers direct); Descriptors(i) materialises file i the way TSampleAnalyser::Analyze returns it.
struct TRecordBatch {
  int mStride = 0;
  int mOffsets[32] = {}, mWidths[32] = {};          // per series, afx_batch_record_layout
  const double* mpRecords = nullptr;                // [frames][stride]
  const double* mpStatistics = nullptr;             // [files][stride][13]
  std::vector<int64_t> mFrameOffset;                // [files + 1]
  std::vector<int32_t> mStatus;                     // [files]: AFX_OK or the per-buffer error
  std::vector<double> mEffectiveLength;             // [files][3]
  std::vector<TSampleDataInfo> mInfo;               // [files]
  // rhythm tracker (afx_batch_fetch_rhythm), in the caller's rhythm buffer
  std::vector<int64_t> mRhythmOffset;               // [files + 1]: rows of the 512/128 frames
  const double* mpRhythmOnsets = nullptr;           // [rows][2]: complex, percussive
  const double* mpRhythmScalars = nullptr;          // [files][14], AFX_R_* order
  const double* mpRhythmStatistics = nullptr;       // [files][2][13]
  double mSeconds[3] = {0, 0, 0};                   // wall time of upload + LoadSample, kernels enqueue, download + wait
  int NumberOfFiles() const { return (int)mStatus.size(); }
  TSampleDescriptors Descriptors(int FileIndex) const;
};

class TSampleAnalyser {
public:
  // Device: HIP device ordinal.  FrameKernel: afx_plan_desc.frame_kernel -- the layout of the STFT kernel, PINNED by
  // default (1 = AFX_FRAME_KERNEL_WAVE64) so that a file's descriptors do not depend on the batch it was analysed in
  // (0 = AFX_FRAME_KERNEL_AUTO lets the library choose by batch size: fastest per batch, not batch-independent;
  // 2 = AFX_FRAME_KERNEL_HALFWAVE).  Throws TReadableException when the GPU path cannot be set up.
  TSampleAnalyser(int SampleRate, int FftFrameSize, int HopFrameSize, int Device = 0, int FrameKernel = 1);
  ~TSampleAnalyser();
  TSampleAnalyser(const TSampleAnalyser&) = delete;
  TSampleAnalyser& operator=(const TSampleAnalyser&) = delete;

  // how the host threads of this analyser's batches wait for the device: sleeping between polls (true) or spinning
  // (the HIP runtime's default); afx_plan_set_blocking_wait
  void SetSleepingWaits(bool Sleeping);

  // frames the loop yields for a normalised buffer (SampleAnalyser.cpp:760-764, 814)
  int64_t NumberOfFrames(int64_t NumberOfSamples) const;

  // AnalyzeLowLevelDescriptors + CalcStatistics (SampleAnalyser.cpp:723-1065) for one decoded, mono,
  // peak-normalised sample (TSampleData::mData); const and thread-safe like the reference
  TSampleDescriptors AnalyzeLowLevelDescriptors(const std::vector<double>& SampleData, bool WithMagnitudes = false) const;

  // the same for many samples in one GPU batch; Failed[i] receives the message for buffers
  // that could not be analysed (one bad sample does not fail the batch, SampleAnalyser.cpp:368-408)
  std::vector<TSampleDescriptors> AnalyzeLowLevelDescriptors(
      const std::vector<const std::vector<double>*>& Samples, std::vector<std::string>* pFailed = nullptr) const;

  // LoadSample + AnalyzeLowLevelDescriptors + CalcStatistics for decoded files, everything on the GPU
  std::vector<TSampleDescriptors> Analyze(const std::vector<TDecodedSample>& Files,
                                          std::vector<TSampleDataInfo>* pInfo = nullptr,
                                          std::vector<std::string>* pFailed = nullptr) const;

  // LoadSample + descriptors + statistics for decoded files, results left as raw records in caller memory:
  // pRecords must hold RecordCapacity doubles, pStatistics Files.size() * 134 * 13 (kMaxStride columns), pRhythm
  // RhythmCapacity doubles (RhythmDoubles() of the files is always enough).  Returns false (and leaves Batch empty)
  // when the records do not fit RecordCapacity or the rhythm results RhythmCapacity: call again with larger buffers.
  enum { kMaxStride = 134 };
  size_t RhythmDoubles(const std::vector<TDecodedSample>& Files) const;
  // does the analyser's device still answer (afx_plan_probe_device)?  false after a fault that took the context down:
  // nothing more can be analysed, as opposed to a batch that failed for its own reasons (memory, a bad file)
  bool DeviceUsable() const;
  static int64_t ConvertedSampleFrames(const TDecodedSample& File, int Rate);   // sample frames once the file is at Rate
  bool AnalyzeToRecords(const std::vector<TDecodedSample>& Files, double* pRecords, size_t RecordCapacity,
                        double* pStatistics, double* pRhythm, size_t RhythmCapacity, TRecordBatch& Batch) const;

private:
  afx_plan* mpPlan;
  int mSampleRate, mFftFrameSize, mHopFrameSize;
};

}  // namespace afec
