// tests/host/test_sample_analyser.cpp -- C++ tests of the host layer, written like the reference's
// own Boost tests (Source/Crawler/FeatureExtraction/Test/TestStatistics.cpp).
//
//   host_test nodevice  error behaviour without a usable device (no GPU needed)
//   host_test columns <out.bin>   column names + encodings of a synthetic TSampleDescriptors (no GPU); the Python
//                                 side compares them with the reference's msgpack output and column list
//   host_test sqlite <db>         writes a synthetic sample and a failed sample into a descriptor database (no GPU)
//   host_test analyse   TSampleAnalyser::AnalyzeLowLevelDescriptors vs the oracle (GPU)
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <random>
#include <vector>

#include "../../afec_amd/host/DescriptorColumns.h"
#include "../../afec_amd/host/SampleAnalyser.h"
#include "../../afec_amd/host/SqlitePool.h"
#include "../../oracle/afx_oracle.h"

static int gFailures = 0;
#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++gFailures; } \
  } while (0)
#define CHECK_EQUAL_EPSILON(a, b, eps) CHECK(std::fabs((a) - (b)) <= (eps))

// the host layer computes nothing itself; without a GPU only its error behaviour can be exercised
static int TestNoDevice() {
  bool Thrown = false;
  try { afec::TSampleAnalyser Analyser(44100, 2048, 1024, /*Device*/ 9999); } catch (const afec::TReadableException&) { Thrown = true; }
  CHECK(Thrown);     // no such device: the constructor reports it like the reference's would (exception), no fallback
  return gFailures;
}

// the same exactly representable values tests/golden/make_golden.py column_values() produces
static std::vector<double> ColumnValues(size_t n, size_t width, long long salt) {
  const size_t m = n * (width ? width : 1);
  std::vector<double> v(m);
  for (size_t i = 0; i < m; ++i) v[i] = (double)(((long long)(i * i) + 7 * salt) % 97) / 8.0 - 3.0;
  if (m > 2) { v[1] = 0.0; v[2] = -0.0; }
  if (m > 3) v[3] = 43.0 * 1000;
  return v;
}

static afec::TSampleDescriptors SyntheticDescriptors() {
  afec::TSampleDescriptors D;
  const size_t Frames = 5;
  D.mEffectiveLength48dB = 1.5; D.mEffectiveLength24dB = 1.25; D.mEffectiveLength12dB = 0.5;
  D.mSpectralCentroid.mValues = ColumnValues(Frames, 0, 1);
  D.mSpectralCentroid.mMedian = 2.5;
  D.mSpectralCentroid.mDVariance = -0.125;
  D.mF0.mValues = ColumnValues(Frames, 0, 2);
  const auto c = ColumnValues(Frames, 14, 3);
  D.mCepstrumBands.mValues.resize(Frames);
  for (size_t fr = 0; fr < Frames; ++fr)
    for (size_t b = 0; b < 14; ++b) D.mCepstrumBands.mValues[fr][b] = c[fr * 14 + b];
  for (size_t b = 0; b < 14; ++b) D.mCepstrumBands.mMean[b] = (double)b / 4.0;
  D.mRhythmComplexOnsets.mValues = ColumnValues(7, 0, 4);
  D.mRhythmComplexOnsets.mMax = 3.25;
  D.mRhythmPercussiveOnsetCount = 4.0;
  D.mRhythmComplexOnsetContrast = -0.125;
  D.mRhythmFinalTempo = 123.5;
  D.mRhythmFinalTempoConfidence = 0.75;
  return D;
}

static void Put(FILE* f, const std::vector<uint8_t>& b) {
  const uint64_t n = b.size();
  std::fwrite(&n, 8, 1, f);
  if (n) std::fwrite(b.data(), 1, n, f);
}

// file: blobs of the golden cases, then the columns of a synthetic sample as (name, type, payload)
static int DumpColumns(const char* pPath) {
  FILE* f = std::fopen(pPath, "wb");
  if (!f) return 1;
  for (size_t n : {0u, 1u, 15u, 16u, 860u, 70000u}) {
    const auto v = ColumnValues(n, 0, n == 70000u ? 5 : (long long)n);
    Put(f, afec::ToMsgpack(v.data(), n));
  }
  const size_t Shapes[4][2] = {{0, 14}, {3, 14}, {20, 28}, {860, 14}};
  for (const auto& s : Shapes) {
    const auto v = ColumnValues(s[0], s[1], (long long)(s[0] + s[1]));
    Put(f, afec::ToMsgpack(v.data(), s[0], s[1]));
  }
  const afec::TSampleDescriptors D = SyntheticDescriptors();
  afec::TSampleDataInfo Info = {0.5f, 0.25f, -2205, 90000};
  const auto Columns = afec::LowLevelColumns(D, &Info);
  const uint64_t Count = Columns.size();
  std::fwrite(&Count, 8, 1, f);
  for (const auto& Col : Columns) {
    const uint64_t Len = Col.mName.size();
    std::fwrite(&Len, 8, 1, f);
    std::fwrite(Col.mName.data(), 1, Len, f);
    const uint8_t Type = (uint8_t)Col.mType;
    std::fwrite(&Type, 1, 1, f);
    if (Col.mType == afec::TColumn::kReal) std::fwrite(&Col.mReal, 8, 1, f);
    else Put(f, Col.mBlob);
  }
  std::fclose(f);
  return 0;
}

static int WriteDatabase(const char* pPath) {
  afec::TSqliteSampleDescriptorPool Pool(pPath);
  const afec::TSampleDescriptors D = SyntheticDescriptors();
  const afec::TSampleDataInfo Info = {0.5f, 0.25f, -2205, 90000};
  afec::TFileProperties File;
  File.mFileType = "wav"; File.mFileSize = 176444; File.mFileLength = 2.0; File.mFileSampleRate = 44100;
  File.mFileChannelCount = 1; File.mFileBitDepth = 16;
  Pool.InsertSample("Kicks/one.wav", 1700000000, File, D, &Info);
  Pool.InsertSample("Kicks/one.wav", 1700000001, File, D, &Info);      // INSERT OR REPLACE: still one row
  Pool.InsertFailedSample("Kicks/broken.wav", 1700000002, "could not decode");
  // the writer's fast path: several files in one transaction, the column vector refilled in place -- a longer and a
  // shorter series than the file before (BLOB buffers are reused), without the load info (one column fewer)
  Pool.BeginTransaction();
  afec::TSampleDescriptors Long = D, Short = D;
  Long.mSpectralCentroid.mValues.assign(40, 0.5);
  Long.mCepstrumBands.mValues.assign(40, std::array<double, 14>{});
  Short.mSpectralCentroid.mValues.assign(1, 7.0);
  Short.mCepstrumBands.mValues.clear();
  Pool.InsertSample("Batch/long.wav", 1700000003, File, Long, &Info);
  Pool.InsertSample("Batch/short.wav", 1700000004, File, Short, &Info);
  Pool.InsertFailedSample("Batch/broken.wav", 1700000005, "could not decode");
  Pool.InsertSample("Batch/noinfo.wav", 1700000006, File, D, nullptr);
  Pool.InsertSample("Batch/again.wav", 1700000007, File, D, &Info);
  // the two-step form the crawler's workers / writer use: values built elsewhere, bound here -- the same row
  {
    std::vector<afec::TColumn> Values;
    afec::RefillLowLevelColumns(Values, Long, &Info);      // a buffer that held another file before
    afec::RefillLowLevelColumns(Values, D, &Info);
    Pool.InsertColumns("Batch/prepared.wav", 1700000007, File, Values);
  }
  Pool.CommitTransaction();
  return 0;
}

// rows per second of the writer alone (no GPU): argv[2] = database path, argv[3] = rows, argv[4] = rows per transaction
// (0: one transaction per row, the reference's way)
static int WriterRate(const char* pPath, int Rows, int PerTransaction, const char* pPragmas) {
  afec::TSqliteSampleDescriptorPool Pool(pPath, pPragmas ? pPragmas : "");
  afec::TSampleDescriptors D = SyntheticDescriptors();
  // a one-second file's worth of frames in every series
  const size_t Frames = 40;
  for (afec::TFramedScalarData* p : {&D.mAmplitudeSilence, &D.mAmplitudePeak, &D.mAmplitudeRms, &D.mAmplitudeEnvelope, &D.mSpectralRms,
                                     &D.mSpectralCentroid, &D.mSpectralRolloff, &D.mSpectralSpread, &D.mSpectralSkewness,
                                     &D.mSpectralKurtosis, &D.mSpectralFlatness, &D.mSpectralInharmonicity, &D.mSpectralComplexity,
                                     &D.mSpectralContrast, &D.mSpectralFlux, &D.mF0, &D.mF0Confidence, &D.mFailSafeF0,
                                     &D.mTristimulus1, &D.mTristimulus2, &D.mTristimulus3, &D.mAutoCorrelation})
    p->mValues.assign(Frames, 0.125);
  D.mRhythmComplexOnsets.mValues.assign(340, 0.0);
  D.mRhythmPercussiveOnsets.mValues.assign(340, 0.0);
  for (afec::TFramedVectorData<14>* p : {&D.mSpectralRmsBands, &D.mSpectralFlatnessBands, &D.mSpectralFluxBands,
                                         &D.mSpectralComplexityBands, &D.mSpectralContrastBands, &D.mCepstrumBands})
    p->mValues.assign(Frames, std::array<double, 14>{});
  D.mSpectrumBands.mValues.assign(Frames, std::array<double, 28>{});
  const afec::TSampleDataInfo Info = {0.5f, 0.25f, -2205, 90000};
  afec::TFileProperties File;
  File.mFileType = "wav"; File.mFileSize = 176444; File.mFileLength = 1.0; File.mFileSampleRate = 44100;
  File.mFileChannelCount = 2; File.mFileBitDepth = 16;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < Rows; ++i) {
    if (PerTransaction > 0 && i % PerTransaction == 0) Pool.BeginTransaction();
    Pool.InsertSample("Rate/file" + std::to_string(i) + ".wav", 1700000000 + i, File, D, &Info);
    if (PerTransaction > 0 && (i % PerTransaction == PerTransaction - 1 || i == Rows - 1)) Pool.CommitTransaction();
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%d rows in %.3f s = %.0f rows/s (%d per transaction)\n", Rows, s, Rows / s, PerTransaction);
  return 0;
}

static int TestAnalyse() {
  std::mt19937 gen(9);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> a(2048 + 1024 * 20), b(5000), c(100);
  for (size_t i = 0; i < a.size(); ++i) a[i] = 0.5 * std::sin(2 * M_PI * 440.0 * i / 44100.0) + 0.1 * U(gen);
  for (auto& v : b) v = U(gen);
  for (auto& v : c) v = U(gen);

  afec::TSampleAnalyser Analyser(44100, 2048, 1024);
  CHECK(Analyser.NumberOfFrames(882000) == 860 && Analyser.NumberOfFrames(900000) == 860);  // 20 s cap
  std::vector<std::string> Failed;
  const auto Results = Analyser.AnalyzeLowLevelDescriptors({&a, &b, &c}, &Failed);
  CHECK(Results.size() == 3 && Failed[0].empty() && Failed[1].empty() && Failed[2].empty());
  CHECK(Results[2].mSpectralRms.mValues.empty());   // shorter than one frame: no frames, no error

  afx_oracle* o = afx_oracle_create(44100, 2048, 1024);
  const std::vector<double>* Inputs[2] = {&a, &b};
  for (int i = 0; i < 2; ++i) {
    const int64_t nf = afx_oracle_num_frames(o, (int64_t)Inputs[i]->size(), 1);
    std::vector<double> rec((size_t)nf * AFXO_RECORD);
    afx_oracle_run(o, Inputs[i]->data(), (int64_t)Inputs[i]->size(), 1, rec.data());
    const auto& R = Results[i];
    CHECK((int64_t)R.mCepstrumBands.mValues.size() == nf);
    for (int64_t f = 0; f < nf; ++f) {
      const double* r = &rec[(size_t)f * AFXO_RECORD];
      for (int k = 0; k < 14; ++k) CHECK_EQUAL_EPSILON(R.mCepstrumBands.mValues[f][k], r[AFXO_MFCC + k], 1e-4 * std::fabs(r[AFXO_MFCC + k]) + 1e-6);
      for (int k = 0; k < 28; ++k) CHECK_EQUAL_EPSILON(R.mSpectrumBands.mValues[f][k], r[AFXO_BANDS + k], 1e-4 * r[AFXO_BANDS + k] + 1e-18);
      for (int k = 0; k < 14; ++k) CHECK_EQUAL_EPSILON(R.mSpectralContrastBands.mValues[f][k], r[AFXO_SUB_CONTRAST + k], 1e-4 * std::fabs(r[AFXO_SUB_CONTRAST + k]) + 1e-9);
      CHECK_EQUAL_EPSILON(R.mSpectralCentroid.mValues[f], r[AFXO_CENTROID], 1e-4 * r[AFXO_CENTROID] + 1e-7);
      CHECK(R.mSpectralRolloff.mValues[f] == r[AFXO_ROLLOFF]);
      CHECK_EQUAL_EPSILON(R.mSpectralFlux.mValues[f], r[AFXO_FLUX], 1e-4 * std::fabs(r[AFXO_FLUX]) + 1e-7);
      CHECK(R.mAmplitudePeak.mValues[f] == r[AFXO_AMP_PEAK]);
    }
    {
      double eff[3];
      afx_oracle_effective_length(o, Inputs[i]->data(), (int64_t)Inputs[i]->size(), eff);
      CHECK(R.mEffectiveLength48dB == eff[0] && R.mEffectiveLength24dB == eff[1] && R.mEffectiveLength12dB == eff[2]);
    }
    // the loop's neighbours
    {
      std::vector<double> nrec((size_t)nf * AFXN_RECORD);
      afx_oracle_run_neighbours(o, Inputs[i]->data(), (int64_t)Inputs[i]->size(), 1, nrec.data());
      CHECK((int64_t)R.mF0.mValues.size() == nf);
      for (int64_t f = 0; f < nf; ++f) {
        const double* r = &nrec[(size_t)f * AFXN_RECORD];
        CHECK(R.mAmplitudeSilence.mValues[f] == r[AFXN_SILENCE]);
        CHECK_EQUAL_EPSILON(R.mAmplitudeEnvelope.mValues[f], r[AFXN_ENVELOPE], 1e-9 * r[AFXN_ENVELOPE]);
        CHECK_EQUAL_EPSILON(R.mF0.mValues[f], r[AFXN_F0], 1e-6 * r[AFXN_F0] + 1e-9);
        CHECK_EQUAL_EPSILON(R.mF0Confidence.mValues[f], r[AFXN_F0_CONF], 1e-6);
        CHECK_EQUAL_EPSILON(R.mFailSafeF0.mValues[f], r[AFXN_F0_FAILSAFE], 1e-6 * r[AFXN_F0_FAILSAFE] + 1e-9);
        CHECK_EQUAL_EPSILON(R.mAutoCorrelation.mValues[f], r[AFXN_AUTOCORR], 1e-8);
        CHECK(R.mSpectralComplexity.mValues[f] == r[AFXN_COMPLEXITY]);
        CHECK(R.mSpectralInharmonicity.mValues[f] == 0.0 && R.mTristimulus1.mValues[f] == 0.0 &&
              R.mTristimulus2.mValues[f] == 0.0 && R.mTristimulus3.mValues[f] == 0.0);
      }
    }
    // per-file statistics of one series against the oracle's TStatistics::Calc
    std::vector<double> series((size_t)nf);
    for (int64_t f = 0; f < nf; ++f) series[(size_t)f] = rec[(size_t)f * AFXO_RECORD + AFXO_CENTROID];
    double s[13] = {0};
    afx_oracle_calc_statistics(series.data(), (int)nf, s);
    CHECK_EQUAL_EPSILON(R.mSpectralCentroid.mMedian, s[2], 1e-4 * s[2]);
    CHECK_EQUAL_EPSILON(R.mSpectralCentroid.mVariance, s[5], 1e-3 * s[5] + 1e-9);
    // statistics of a vector series and of a neighbour: the GPU's reduction of its own series
    for (int k : {0, 5, 13}) {
      for (int64_t f = 0; f < nf; ++f) series[(size_t)f] = R.mCepstrumBands.mValues[(size_t)f][k];
      afx_oracle_calc_statistics(series.data(), (int)nf, s);
      CHECK_EQUAL_EPSILON(R.mCepstrumBands.mMean[k], s[3], 1e-9 * std::fabs(s[3]) + 1e-12);
      CHECK_EQUAL_EPSILON(R.mCepstrumBands.mMedian[k], s[2], 1e-12);
      CHECK_EQUAL_EPSILON(R.mCepstrumBands.mDVariance[k], s[12], 1e-9 * std::fabs(s[12]) + 1e-12);
    }
    afx_oracle_calc_statistics(R.mF0.mValues.data(), (int)nf, s);
    CHECK_EQUAL_EPSILON(R.mF0.mMax, s[1], 0.0);
    CHECK_EQUAL_EPSILON(R.mF0.mMean, s[3], 1e-9 * std::fabs(s[3]) + 1e-12);
    // the rhythm tracker (SampleAnalyser.cpp:983-1048): onsets of both functions, the 14 scalars, onset statistics
    {
      const int64_t nt = afx_oracle_rhythm_frames(o, (int64_t)Inputs[i]->size(), 1);
      std::vector<double> onsets((size_t)nt * 2);
      double sc[14];
      afx_oracle_run_rhythm(o, Inputs[i]->data(), (int64_t)Inputs[i]->size(), 1, 44100, (int64_t)Inputs[i]->size(), 0,
                            onsets.data(), nullptr, nullptr, sc);
      CHECK((int64_t)R.mRhythmComplexOnsets.mValues.size() == nt && (int64_t)R.mRhythmPercussiveOnsets.mValues.size() == nt);
      for (int64_t t = 0; t < nt; ++t) {
        CHECK_EQUAL_EPSILON(R.mRhythmComplexOnsets.mValues[(size_t)t], onsets[(size_t)t], 1e-5);
        CHECK_EQUAL_EPSILON(R.mRhythmPercussiveOnsets.mValues[(size_t)t], onsets[(size_t)(nt + t)], 1e-5);
      }
      const double got[14] = {R.mRhythmComplexOnsetCount, R.mRhythmComplexTempo, R.mRhythmComplexTempoConfidence,
                              R.mRhythmComplexOnsetFrequencyMean, R.mRhythmComplexOnsetStrength, R.mRhythmComplexOnsetContrast,
                              R.mRhythmPercussiveOnsetCount, R.mRhythmPercussiveTempo, R.mRhythmPercussiveTempoConfidence,
                              R.mRhythmPercussiveOnsetFrequencyMean, R.mRhythmPercussiveOnsetStrength,
                              R.mRhythmPercussiveOnsetContrast, R.mRhythmFinalTempo, R.mRhythmFinalTempoConfidence};
      for (int k = 0; k < 14; ++k) CHECK_EQUAL_EPSILON(got[k], sc[k], 1e-5 * std::fabs(sc[k]) + 1e-9);
      afx_oracle_calc_statistics(R.mRhythmComplexOnsets.mValues.data(), (int)nt, s);
      CHECK_EQUAL_EPSILON(R.mRhythmComplexOnsets.mMax, s[1], 0.0);
      CHECK_EQUAL_EPSILON(R.mRhythmComplexOnsets.mMean, s[3], 1e-9 * std::fabs(s[3]) + 1e-12);
    }
  }
  afx_oracle_destroy(o);

  // decoded files through the GPU LoadSample front end: same descriptors as analysing the oracle's normalised buffer
  {
    std::vector<short> Pcm(2 * 30000);
    for (int i = 0; i < 30000; ++i) {
      const double v = 0.4 * std::sin(2 * M_PI * 330.0 * i / 44100.0) * std::exp(-i / 9000.0) + 0.01 * U(gen);
      Pcm[2 * i] = (short)std::lround(v * 20000.0);
      Pcm[2 * i + 1] = (short)std::lround(0.7 * v * 20000.0);
    }
    std::vector<afec::TSampleDataInfo> Info;
    std::vector<std::string> LoadFailed;
    const auto FromFile = Analyser.Analyze({{Pcm.data(), /*AFX_RAW_I16*/ 0, 2, 44100, 30000}}, &Info, &LoadFailed);
    CHECK(FromFile.size() == 1 && LoadFailed[0].empty());
    afx_oracle_load_info OracleInfo;
    double* pMono = afx_oracle_load_sample(Pcm.data(), 0, 2, 30000, 2048, &OracleInfo);
    CHECK(Info[0].mNumberOfSamples == OracleInfo.n_samples && Info[0].mDataOffset == OracleInfo.data_offset);
    CHECK(Info[0].mPeakValue == OracleInfo.peak_value);
    const std::vector<double> Mono(pMono, pMono + OracleInfo.n_samples);
    afx_oracle_free(pMono);
    const auto FromBuffer = Analyser.AnalyzeLowLevelDescriptors(Mono);
    CHECK(FromFile[0].mF0.mValues == FromBuffer.mF0.mValues);
    CHECK(FromFile[0].mSpectralCentroid.mValues == FromBuffer.mSpectralCentroid.mValues);
    CHECK(FromFile[0].mCepstrumBands.mMedian == FromBuffer.mCepstrumBands.mMedian);
    CHECK(FromFile[0].mEffectiveLength24dB == FromBuffer.mEffectiveLength24dB);
    CHECK(FromFile[0].mRhythmComplexOnsets.mValues == FromBuffer.mRhythmComplexOnsets.mValues);
    CHECK(FromFile[0].mRhythmPercussiveOnsetContrast == FromBuffer.mRhythmPercussiveOnsetContrast);
  }

  // error behaviour: an unsupported geometry throws like the reference's constructor would assert
  bool Thrown = false;
  try { afec::TSampleAnalyser Bad(48000, 2048, 1024); } catch (const afec::TReadableException&) { Thrown = true; }
  CHECK(Thrown);
  return gFailures;
}

int main(int argc, char** argv) {
  int rc = 2;
  try {
    if (argc >= 2 && !std::strcmp(argv[1], "nodevice")) rc = TestNoDevice();
    else if (argc >= 3 && !std::strcmp(argv[1], "columns")) rc = DumpColumns(argv[2]);
    else if (argc >= 3 && !std::strcmp(argv[1], "sqlite")) rc = WriteDatabase(argv[2]);
    else if (argc >= 5 && !std::strcmp(argv[1], "writer_rate")) rc = WriterRate(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), argc >= 6 ? argv[5] : nullptr);
    else if (argc >= 2 && !std::strcmp(argv[1], "analyse")) rc = TestAnalyse();
  } catch (const std::exception& e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return 3;
  }
  if (rc == 0) std::printf("OK\n");
  return rc;
}
