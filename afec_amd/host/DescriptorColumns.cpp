// afec_amd/host/DescriptorColumns.cpp -- see DescriptorColumns.h.
#include "DescriptorColumns.h"

#include <cstring>

namespace afec {

namespace {

void PutArrayHeader(std::vector<uint8_t>& Out, size_t n) {     // msgpack array family
  if (n < 16) {
    Out.push_back((uint8_t)(0x90u | n));
  } else if (n < 65536) {
    Out.push_back(0xdc);
    Out.push_back((uint8_t)(n >> 8));
    Out.push_back((uint8_t)n);
  } else {
    Out.push_back(0xdd);
    for (int s = 24; s >= 0; s -= 8) Out.push_back((uint8_t)(n >> s));
  }
}

void PutDouble(std::vector<uint8_t>& Out, double v) {           // float 64: 0xcb, big-endian IEEE bits
  uint64_t bits;
  std::memcpy(&bits, &v, 8);
  Out.push_back(0xcb);
  for (int s = 56; s >= 0; s -= 8) Out.push_back((uint8_t)(bits >> s));
}

TColumn Real(const std::string& Name, double v) { return TColumn{Name + "_R", TColumn::kReal, v, {}}; }

// TFramedScalarData::OnValues (Export/SampleDescriptors.h:187-210)
void Append(std::vector<TColumn>& Out, const char* pName, const TFramedScalarData& d) {
  const std::string n(pName);
  Out.push_back(TColumn{n + "_VR", TColumn::kBlob, 0.0, ToMsgpack(d.mValues.data(), d.mValues.size())});
  const std::pair<const char*, double> Stats[] = {
      {"_min", d.mMin}, {"_max", d.mMax}, {"_median", d.mMedian}, {"_mean", d.mMean}, {"_gmean", d.mGeometricMean},
      {"_variance", d.mVariance}, {"_centroid", d.mCentroid}, {"_spread", d.mSpread}, {"_skewness", d.mSkewness},
      {"_kurtosis", d.mKurtosis}, {"_flatness", d.mFlatness}, {"_dmean", d.mDMean}, {"_dvariance", d.mDVariance}};
  for (const auto& s : Stats) Out.push_back(Real(n + s.first, s.second));
}

// TFramedVectorData<W>::OnValues (Export/SampleDescriptors.h:302-325)
template <int W>
void Append(std::vector<TColumn>& Out, const char* pName, const TFramedVectorData<W>& d) {
  const std::string n(pName);
  const double* pValues = d.mValues.empty() ? nullptr : d.mValues[0].data();   // std::array rows are contiguous
  Out.push_back(TColumn{n + "_VVR", TColumn::kBlob, 0.0, ToMsgpack(pValues, d.mValues.size(), (size_t)W)});
  const std::pair<const char*, const std::array<double, W>*> Stats[] = {
      {"_min", &d.mMin}, {"_max", &d.mMax}, {"_median", &d.mMedian}, {"_mean", &d.mMean},
      {"_gmean", &d.mGeometricMean}, {"_variance", &d.mVariance}, {"_centroid", &d.mCentroid},
      {"_spread", &d.mSpread}, {"_skewness", &d.mSkewness}, {"_kurtosis", &d.mKurtosis},
      {"_flatness", &d.mFlatness}, {"_dmean", &d.mDMean}, {"_dvariance", &d.mDVariance}};
  for (const auto& s : Stats)
    Out.push_back(TColumn{n + s.first + "_VR", TColumn::kBlob, 0.0, ToMsgpack(s.second->data(), (size_t)W)});
}

}  // namespace

std::vector<uint8_t> ToMsgpack(const double* pValues, size_t Count) {
  std::vector<uint8_t> Out;
  Out.reserve(5 + 9 * Count);
  PutArrayHeader(Out, Count);
  for (size_t i = 0; i < Count; ++i) PutDouble(Out, pValues[i]);
  return Out;
}

std::vector<uint8_t> ToMsgpack(const double* pValues, size_t Rows, size_t Width) {
  std::vector<uint8_t> Out;
  Out.reserve(5 + Rows * (3 + 9 * Width));
  PutArrayHeader(Out, Rows);
  for (size_t r = 0; r < Rows; ++r) {
    PutArrayHeader(Out, Width);
    for (size_t j = 0; j < Width; ++j) PutDouble(Out, pValues[r * Width + j]);
  }
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
  // order of TSampleDescriptors::Descriptors(kLowLevelDescriptors), SampleDescriptors.cpp:150-205 (the file_*
  // descriptors in front come from the container, not from this library)
  Out.push_back(Real("effectve_length_48dB", D.mEffectiveLength48dB));   // [sic], SampleDescriptors.cpp:40-42
  Out.push_back(Real("effectve_length_24dB", D.mEffectiveLength24dB));
  Out.push_back(Real("effectve_length_12dB", D.mEffectiveLength12dB));
  if (pInfo) {
    // TAudioMath::SamplesToMs(rate, mDataOffset) / 1000.0 with SamplesToMs in float (SampleAnalyser.cpp:748-749)
    const float Ms = (float)pInfo->mDataOffset / ((float)SampleRate / 1000.0f);
    Out.push_back(Real("analyzation_offset", (double)Ms / 1000.0));
  }
  Append(Out, "amplitude_silence", D.mAmplitudeSilence);
  Append(Out, "amplitude_peak", D.mAmplitudePeak);
  Append(Out, "amplitude_rms", D.mAmplitudeRms);
  Append(Out, "amplitude_envelope", D.mAmplitudeEnvelope);
  Append(Out, "spectral_rms", D.mSpectralRms);
  Append(Out, "spectral_centroid", D.mSpectralCentroid);
  Append(Out, "spectral_rolloff", D.mSpectralRolloff);
  Append(Out, "spectral_spread", D.mSpectralSpread);
  Append(Out, "spectral_skewness", D.mSpectralSkewness);
  Append(Out, "spectral_kurtosis", D.mSpectralKurtosis);
  Append(Out, "spectral_flatness", D.mSpectralFlatness);
  Append(Out, "spectral_inharmonicity", D.mSpectralInharmonicity);
  Append(Out, "spectral_complexity", D.mSpectralComplexity);
  Append(Out, "spectral_contrast", D.mSpectralContrast);
  Append(Out, "spectral_flux", D.mSpectralFlux);
  Append(Out, "f0", D.mF0);
  Append(Out, "f0_confidence", D.mF0Confidence);
  Append(Out, "failsafe_f0", D.mFailSafeF0);
  Append(Out, "tristimulus1", D.mTristimulus1);
  Append(Out, "tristimulus2", D.mTristimulus2);
  Append(Out, "tristimulus3", D.mTristimulus3);
  Append(Out, "auto_correlation", D.mAutoCorrelation);
  // rhythm tracker, SampleDescriptors.cpp:180-195
  Append(Out, "rhythm_complex_onsets", D.mRhythmComplexOnsets);
  Out.push_back(Real("rhythm_complex_onset_count", D.mRhythmComplexOnsetCount));
  Out.push_back(Real("rhythm_complex_onset_contrast", D.mRhythmComplexOnsetContrast));
  Out.push_back(Real("rhythm_complex_onset_frequency_mean", D.mRhythmComplexOnsetFrequencyMean));
  Out.push_back(Real("rhythm_complex_onset_strength", D.mRhythmComplexOnsetStrength));
  Out.push_back(Real("rhythm_complex_tempo", D.mRhythmComplexTempo));
  Out.push_back(Real("rhythm_complex_tempo_confidence", D.mRhythmComplexTempoConfidence));
  Append(Out, "rhythm_percussive_onsets", D.mRhythmPercussiveOnsets);
  Out.push_back(Real("rhythm_percussive_onset_count", D.mRhythmPercussiveOnsetCount));
  Out.push_back(Real("rhythm_percussive_onset_contrast", D.mRhythmPercussiveOnsetContrast));
  Out.push_back(Real("rhythm_percussive_onset_frequency_mean", D.mRhythmPercussiveOnsetFrequencyMean));
  Out.push_back(Real("rhythm_percussive_onset_strength", D.mRhythmPercussiveOnsetStrength));
  Out.push_back(Real("rhythm_percussive_tempo", D.mRhythmPercussiveTempo));
  Out.push_back(Real("rhythm_percussive_tempo_confidence", D.mRhythmPercussiveTempoConfidence));
  Out.push_back(Real("rhythm_final_tempo", D.mRhythmFinalTempo));
  Out.push_back(Real("rhythm_final_tempo_confidence", D.mRhythmFinalTempoConfidence));
  Append(Out, "spectral_rms_bands", D.mSpectralRmsBands);
  Append(Out, "spectral_flatness_bands", D.mSpectralFlatnessBands);
  Append(Out, "spectral_flux_bands", D.mSpectralFluxBands);
  Append(Out, "spectral_complexity_bands", D.mSpectralComplexityBands);
  Append(Out, "spectral_contrast_bands", D.mSpectralContrastBands);
  Append(Out, "frequency_bands", D.mSpectrumBands);
  Append(Out, "cepstrum_bands", D.mCepstrumBands);
  return Out;
}

}  // namespace afec
