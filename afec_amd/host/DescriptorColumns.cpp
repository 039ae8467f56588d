// afec_amd/host/DescriptorColumns.cpp -- see DescriptorColumns.h.
#include "DescriptorColumns.h"

#include <cstring>

namespace afec {

namespace {

const char* const kStatPostfix[13] = {"_min", "_max", "_median", "_mean", "_gmean", "_variance", "_centroid", "_spread",
                                      "_skewness", "_kurtosis", "_flatness", "_dmean", "_dvariance"};

// msgpack into a vector that keeps its capacity between files: the size is known up front, the bytes are written
// through a pointer (a one-second file has ~6 000 doubles in its row; byte-wise push_back was the writer's largest cost)
inline size_t HeaderBytes(size_t n) { return n < 16 ? 1 : (n < 65536 ? 3 : 5); }
inline uint8_t* WriteHeader(uint8_t* p, size_t n) {
  if (n < 16) { *p++ = (uint8_t)(0x90u | n); }
  else if (n < 65536) { *p++ = 0xdc; *p++ = (uint8_t)(n >> 8); *p++ = (uint8_t)n; }
  else { *p++ = 0xdd; for (int s = 24; s >= 0; s -= 8) *p++ = (uint8_t)(n >> s); }
  return p;
}
inline uint8_t* WriteDouble(uint8_t* p, double v) {
  uint64_t bits;
  std::memcpy(&bits, &v, 8);
  bits = __builtin_bswap64(bits);      // big-endian IEEE bits behind 0xcb
  *p++ = 0xcb;
  std::memcpy(p, &bits, 8);
  return p + 8;
}
void PackInto(std::vector<uint8_t>& Out, const double* pValues, size_t Count) {
  Out.resize(HeaderBytes(Count) + 9 * Count);
  uint8_t* p = WriteHeader(Out.data(), Count);
  for (size_t i = 0; i < Count; ++i) p = WriteDouble(p, pValues[i]);
}
void PackInto(std::vector<uint8_t>& Out, const double* pValues, size_t Rows, size_t Width) {
  Out.resize(HeaderBytes(Rows) + Rows * (HeaderBytes(Width) + 9 * Width));
  uint8_t* p = WriteHeader(Out.data(), Rows);
  for (size_t r = 0; r < Rows; ++r) {
    p = WriteHeader(p, Width);
    for (size_t j = 0; j < Width; ++j) p = WriteDouble(p, pValues[r * Width + j]);
  }
}

// Walks the columns in the order of TSampleDescriptors::Descriptors(kLowLevelDescriptors).  With Named the columns are
// created (names built: ~460 strings); without, an existing vector from an earlier walk is refilled in place -- same
// order, so only the values and the BLOB contents change and nothing is allocated (the writer's per-file cost).
template <bool Named>
struct TColumnWalk {
  std::vector<TColumn>& mOut;
  size_t mNext = 0;
  TColumn& Next(const char* pBase, const char* pMiddle, const char* pPostfix, TColumn::TType Type) {
    if (Named) {
      mOut.push_back(TColumn{std::string(pBase) + pMiddle + pPostfix, Type, 0.0, {}});
      return mOut.back();
    }
    return mOut[mNext++];
  }
  void Real(const char* pName, double v) { Next(pName, "", "_R", TColumn::kReal).mReal = v; }
  // TFramedScalarData::OnValues (Export/SampleDescriptors.h:187-210)
  void Series(const char* pName, const TFramedScalarData& d) {
    PackInto(Next(pName, "", "_VR", TColumn::kBlob).mBlob, d.mValues.data(), d.mValues.size());
    const double Stats[13] = {d.mMin, d.mMax, d.mMedian, d.mMean, d.mGeometricMean, d.mVariance, d.mCentroid, d.mSpread,
                              d.mSkewness, d.mKurtosis, d.mFlatness, d.mDMean, d.mDVariance};
    for (int i = 0; i < 13; ++i) Next(pName, kStatPostfix[i], "_R", TColumn::kReal).mReal = Stats[i];
  }
  // TFramedVectorData<W>::OnValues (Export/SampleDescriptors.h:302-325)
  template <int W>
  void Series(const char* pName, const TFramedVectorData<W>& d) {
    const double* pValues = d.mValues.empty() ? nullptr : d.mValues[0].data();   // std::array rows are contiguous
    PackInto(Next(pName, "", "_VVR", TColumn::kBlob).mBlob, pValues, d.mValues.size(), (size_t)W);
    const std::array<double, W>* Stats[13] = {&d.mMin, &d.mMax, &d.mMedian, &d.mMean, &d.mGeometricMean, &d.mVariance,
                                              &d.mCentroid, &d.mSpread, &d.mSkewness, &d.mKurtosis, &d.mFlatness,
                                              &d.mDMean, &d.mDVariance};
    for (int i = 0; i < 13; ++i) PackInto(Next(pName, kStatPostfix[i], "_VR", TColumn::kBlob).mBlob, Stats[i]->data(), (size_t)W);
  }
};

template <bool Named>
void WalkLowLevel(std::vector<TColumn>& Out, const TSampleDescriptors& D, const TSampleDataInfo* pInfo, int SampleRate) {
  TColumnWalk<Named> w{Out};
  // order of TSampleDescriptors::Descriptors(kLowLevelDescriptors), SampleDescriptors.cpp:150-205 (the file_*
  // descriptors in front come from the container, not from this library)
  w.Real("effectve_length_48dB", D.mEffectiveLength48dB);   // [sic], SampleDescriptors.cpp:40-42
  w.Real("effectve_length_24dB", D.mEffectiveLength24dB);
  w.Real("effectve_length_12dB", D.mEffectiveLength12dB);
  if (pInfo) {
    // TAudioMath::SamplesToMs(rate, mDataOffset) / 1000.0 with SamplesToMs in float (SampleAnalyser.cpp:748-749)
    const float Ms = (float)pInfo->mDataOffset / ((float)SampleRate / 1000.0f);
    w.Real("analyzation_offset", (double)Ms / 1000.0);
  }
  w.Series("amplitude_silence", D.mAmplitudeSilence);
  w.Series("amplitude_peak", D.mAmplitudePeak);
  w.Series("amplitude_rms", D.mAmplitudeRms);
  w.Series("amplitude_envelope", D.mAmplitudeEnvelope);
  w.Series("spectral_rms", D.mSpectralRms);
  w.Series("spectral_centroid", D.mSpectralCentroid);
  w.Series("spectral_rolloff", D.mSpectralRolloff);
  w.Series("spectral_spread", D.mSpectralSpread);
  w.Series("spectral_skewness", D.mSpectralSkewness);
  w.Series("spectral_kurtosis", D.mSpectralKurtosis);
  w.Series("spectral_flatness", D.mSpectralFlatness);
  w.Series("spectral_inharmonicity", D.mSpectralInharmonicity);
  w.Series("spectral_complexity", D.mSpectralComplexity);
  w.Series("spectral_contrast", D.mSpectralContrast);
  w.Series("spectral_flux", D.mSpectralFlux);
  w.Series("f0", D.mF0);
  w.Series("f0_confidence", D.mF0Confidence);
  w.Series("failsafe_f0", D.mFailSafeF0);
  w.Series("tristimulus1", D.mTristimulus1);
  w.Series("tristimulus2", D.mTristimulus2);
  w.Series("tristimulus3", D.mTristimulus3);
  w.Series("auto_correlation", D.mAutoCorrelation);
  // rhythm tracker, SampleDescriptors.cpp:180-195
  w.Series("rhythm_complex_onsets", D.mRhythmComplexOnsets);
  w.Real("rhythm_complex_onset_count", D.mRhythmComplexOnsetCount);
  w.Real("rhythm_complex_onset_contrast", D.mRhythmComplexOnsetContrast);
  w.Real("rhythm_complex_onset_frequency_mean", D.mRhythmComplexOnsetFrequencyMean);
  w.Real("rhythm_complex_onset_strength", D.mRhythmComplexOnsetStrength);
  w.Real("rhythm_complex_tempo", D.mRhythmComplexTempo);
  w.Real("rhythm_complex_tempo_confidence", D.mRhythmComplexTempoConfidence);
  w.Series("rhythm_percussive_onsets", D.mRhythmPercussiveOnsets);
  w.Real("rhythm_percussive_onset_count", D.mRhythmPercussiveOnsetCount);
  w.Real("rhythm_percussive_onset_contrast", D.mRhythmPercussiveOnsetContrast);
  w.Real("rhythm_percussive_onset_frequency_mean", D.mRhythmPercussiveOnsetFrequencyMean);
  w.Real("rhythm_percussive_onset_strength", D.mRhythmPercussiveOnsetStrength);
  w.Real("rhythm_percussive_tempo", D.mRhythmPercussiveTempo);
  w.Real("rhythm_percussive_tempo_confidence", D.mRhythmPercussiveTempoConfidence);
  w.Real("rhythm_final_tempo", D.mRhythmFinalTempo);
  w.Real("rhythm_final_tempo_confidence", D.mRhythmFinalTempoConfidence);
  w.Series("spectral_rms_bands", D.mSpectralRmsBands);
  w.Series("spectral_flatness_bands", D.mSpectralFlatnessBands);
  w.Series("spectral_flux_bands", D.mSpectralFluxBands);
  w.Series("spectral_complexity_bands", D.mSpectralComplexityBands);
  w.Series("spectral_contrast_bands", D.mSpectralContrastBands);
  w.Series("frequency_bands", D.mSpectrumBands);
  w.Series("cepstrum_bands", D.mCepstrumBands);
}

}  // namespace

std::vector<uint8_t> ToMsgpack(const double* pValues, size_t Count) {
  std::vector<uint8_t> Out;
  PackInto(Out, pValues, Count);
  return Out;
}

std::vector<uint8_t> ToMsgpack(const double* pValues, size_t Rows, size_t Width) {
  std::vector<uint8_t> Out;
  PackInto(Out, pValues, Rows, Width);
  return Out;
}

std::vector<TColumnSpec> LowLevelSchema() {
  static const char* const kStats[] = {"_min", "_max", "_median", "_mean", "_gmean", "_variance", "_centroid", "_spread",
                                       "_skewness", "_kurtosis", "_flatness", "_dmean", "_dvariance"};
  std::vector<TColumnSpec> Out;
  auto Scalar = [&](const char* pName, const char* pPostfix, const char* pType) {
    Out.push_back(TColumnSpec{std::string(pName) + "_" + pPostfix, pType});
  };
  auto FramedScalar = [&](const char* pName) {      // TFramedScalarData: VR BLOB + 13 REAL
    Out.push_back(TColumnSpec{std::string(pName) + "_VR", "BLOB"});
    for (const char* s : kStats) Out.push_back(TColumnSpec{std::string(pName) + s + "_R", "REAL"});
  };
  auto FramedVector = [&](const char* pName) {      // TFramedVectorData<W>: VVR BLOB + 13 VR BLOB
    Out.push_back(TColumnSpec{std::string(pName) + "_VVR", "BLOB"});
    for (const char* s : kStats) Out.push_back(TColumnSpec{std::string(pName) + s + "_VR", "BLOB"});
  };
  // SampleDescriptors.cpp:29-37, 150-157 (shared), Export/SampleDescriptors.h:384-390 (types)
  Scalar("file_type", "S", "TEXT");
  Scalar("file_size", "R", "INTEGER");
  Scalar("file_length", "R", "REAL");
  Scalar("file_sample_rate", "R", "INTEGER");
  Scalar("file_channel_count", "R", "INTEGER");
  Scalar("file_bit_depth", "R", "INTEGER");
  // SampleDescriptors.cpp:159-205 (low level)
  for (const char* n : {"effectve_length_48dB", "effectve_length_24dB", "effectve_length_12dB", "analyzation_offset"})
    Scalar(n, "R", "REAL");
  for (const char* n : {"amplitude_silence", "amplitude_peak", "amplitude_rms", "amplitude_envelope", "spectral_rms",
                        "spectral_centroid", "spectral_rolloff", "spectral_spread", "spectral_skewness",
                        "spectral_kurtosis", "spectral_flatness", "spectral_inharmonicity", "spectral_complexity",
                        "spectral_contrast", "spectral_flux", "f0", "f0_confidence", "failsafe_f0", "tristimulus1",
                        "tristimulus2", "tristimulus3", "auto_correlation"})
    FramedScalar(n);
  for (const char* kind : {"rhythm_complex", "rhythm_percussive"}) {
    const std::string k(kind);
    FramedScalar((k + "_onsets").c_str());
    for (const char* n : {"_onset_count", "_onset_contrast", "_onset_frequency_mean", "_onset_strength", "_tempo",
                          "_tempo_confidence"})
      Scalar((k + n).c_str(), "R", "REAL");
  }
  Scalar("rhythm_final_tempo", "R", "REAL");
  Scalar("rhythm_final_tempo_confidence", "R", "REAL");
  for (const char* n : {"spectral_rms_bands", "spectral_flatness_bands", "spectral_flux_bands",
                        "spectral_complexity_bands", "spectral_contrast_bands", "frequency_bands", "cepstrum_bands"})
    FramedVector(n);
  return Out;
}

std::vector<TColumn> LowLevelColumns(const TSampleDescriptors& D, const TSampleDataInfo* pInfo, int SampleRate) {
  std::vector<TColumn> Out;
  WalkLowLevel<true>(Out, D, pInfo, SampleRate);
  return Out;
}

void RefillLowLevelColumns(std::vector<TColumn>& Columns, const TSampleDescriptors& D, const TSampleDataInfo* pInfo, int SampleRate) {
  // (4 scalars with pInfo, 3 without: the only thing the column count depends on)
  const size_t Expected = 3 + (pInfo ? 1 : 0) + 22 * 14 + 2 * (14 + 6) + 2 + 7 * 14;
  if (Columns.size() != Expected) {
    Columns.clear();
    WalkLowLevel<true>(Columns, D, pInfo, SampleRate);
  } else {
    WalkLowLevel<false>(Columns, D, pInfo, SampleRate);
  }
}

}  // namespace afec
