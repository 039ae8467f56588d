// tests/sanitize/sanitize_main.cpp -- CPU build of the host-side logic and of the oracle under AddressSanitizer +
// UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool; tools/sanitize_cpu.sh builds and runs this).
//   * WAV reader: well-formed files of every sample type, then 20 000 mutated / truncated images (must either parse
//     or throw TReadableException, never read outside the image);
//   * msgpack column encoder on empty / large series;
//   * oracle: per-frame loop, neighbours, LoadSample, statistics, rhythm tracker on ragged and degenerate inputs
//     (the checker must itself be memory-clean, the GPU parity tests trust it).
#include <cmath>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../afec_amd/host/DescriptorColumns.h"
#include "../../afec_amd/host/WaveFile.h"
#include "../../include/afx.h"
extern "C" {
#include "../../oracle/afx_oracle.h"
}

static std::vector<unsigned char> MakeWav(int Channels, int Rate, int Bits, int Format, int Frames, std::mt19937& Gen,
                                          bool ExtraChunk) {
  std::vector<unsigned char> w;
  auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) w.push_back((unsigned char)(v >> (8 * i))); };
  auto put16 = [&](uint16_t v) { w.push_back((unsigned char)v); w.push_back((unsigned char)(v >> 8)); };
  const uint32_t DataBytes = (uint32_t)Frames * Channels * (Bits / 8);
  w.insert(w.end(), {'R', 'I', 'F', 'F'});
  put32(0);
  w.insert(w.end(), {'W', 'A', 'V', 'E'});
  if (ExtraChunk) { w.insert(w.end(), {'L', 'I', 'S', 'T'}); put32(5); for (int i = 0; i < 6; ++i) w.push_back('x'); }
  w.insert(w.end(), {'f', 'm', 't', ' '});
  put32(16);
  put16((uint16_t)Format); put16((uint16_t)Channels); put32((uint32_t)Rate);
  put32((uint32_t)(Rate * Channels * Bits / 8)); put16((uint16_t)(Channels * Bits / 8)); put16((uint16_t)Bits);
  w.insert(w.end(), {'d', 'a', 't', 'a'});
  put32(DataBytes);
  for (uint32_t i = 0; i < DataBytes; ++i) w.push_back((unsigned char)Gen());
  const uint32_t Riff = (uint32_t)w.size() - 8;
  std::memcpy(&w[4], &Riff, 4);
  return w;
}

int main() {
  std::mt19937 Gen(12345);
  int Parsed = 0, Rejected = 0;
  // ---- WAV reader ----
  const int Formats[][2] = {{8, 1}, {16, 1}, {24, 1}, {32, 1}, {32, 3}, {64, 3}};
  for (const auto& f : Formats)
    for (int Channels : {1, 2, 8})
      for (bool Extra : {false, true}) {
        const auto Image = MakeWav(Channels, 44100, f[0], f[1], 777, Gen, Extra);
        afec::TWaveFile Wave;
        Wave.OpenForRead(Image.data(), Image.size());
        std::vector<unsigned char> Storage;
        const afec::TDecodedSample s = Wave.DecodedSample(Storage);
        if (s.mNumberOfSampleFrames != 777 || s.mNumberOfChannels != Channels) { std::printf("bad parse\n"); return 1; }
        // touch every byte the front end would read
        const size_t Bps = s.mFormat == AFX_RAW_I16 ? 2 : (s.mFormat == AFX_RAW_I24 ? 3 : (s.mFormat == AFX_RAW_F64 ? 8 : 4));
        unsigned Sum = 0;
        const unsigned char* p = (const unsigned char*)s.mpInterleavedSamples;
        for (size_t i = 0; i < (size_t)s.mNumberOfSampleFrames * Channels * Bps; ++i) Sum += p[i];
        (void)Sum;
        ++Parsed;
      }
  const auto Base = MakeWav(2, 44100, 16, 1, 300, Gen, true);
  for (int Trial = 0; Trial < 20000; ++Trial) {
    std::vector<unsigned char> Image = Base;
    const int Kind = Trial % 4;
    if (Kind == 0) Image.resize(Gen() % (Image.size() + 1));                       // truncation
    else if (Kind == 1) for (int k = 0; k < 3; ++k) Image[Gen() % 80 % Image.size()] = (unsigned char)Gen();   // header bytes
    else if (Kind == 2) { const uint32_t v = Gen(); std::memcpy(&Image[(Gen() % 19) * 4], &v, 4); }            // size fields
    else { Image.resize(Gen() % 64); for (auto& b : Image) b = (unsigned char)Gen(); }                        // noise
    // exactly-sized heap copy: any read past the end is an ASan error
    unsigned char* Exact = (unsigned char*)std::malloc(Image.size() ? Image.size() : 1);
    std::memcpy(Exact, Image.data(), Image.size());
    try {
      afec::TWaveFile Wave;
      Wave.OpenForRead(Exact, Image.size());
      std::vector<unsigned char> Storage;
      const afec::TDecodedSample s = Wave.DecodedSample(Storage);
      const size_t Bps = s.mFormat == AFX_RAW_I16 ? 2 : (s.mFormat == AFX_RAW_I24 ? 3 : (s.mFormat == AFX_RAW_F64 ? 8 : 4));
      unsigned Sum = 0;
      const unsigned char* p = (const unsigned char*)s.mpInterleavedSamples;
      for (size_t i = 0; i < (size_t)s.mNumberOfSampleFrames * s.mNumberOfChannels * Bps; ++i) Sum += p[i];
      (void)Sum;
      ++Parsed;
    } catch (const afec::TReadableException&) {
      ++Rejected;
    }
    std::free(Exact);
  }
  std::printf("wav: %d parsed, %d rejected\n", Parsed, Rejected);

  // ---- WAV reader on files (head parsed from the first 4 KiB, data chunk by pread): the same mutations, written to disk ----
  {
    int FromDisk = 0, RejectedOnDisk = 0, Differ = 0;
    const std::string Path = "/tmp/afx_san/mutated.wav";
    for (int Trial = 0; Trial < 3000; ++Trial) {
      // every third file has a large chunk in front of the data chunk: the data lies behind the 4 KiB head
      std::vector<unsigned char> Image = MakeWav(1 + Trial % 2, 44100, (Trial % 5 == 0) ? 8 : 16, 1, 50 + Gen() % 4000, Gen, Trial % 3 == 0);
      if (Trial % 3 == 0) {
        std::vector<unsigned char> Big(Image.begin(), Image.begin() + 12);
        const unsigned char Junk[8] = {'J', 'U', 'N', 'K', 0x00, 0x20, 0x00, 0x00};   // 8192 bytes
        Big.insert(Big.end(), Junk, Junk + 8);
        Big.resize(Big.size() + 8192, 0x55);
        Big.insert(Big.end(), Image.begin() + 12, Image.end());
        Image.swap(Big);
      }
      const int Kind = Trial % 4;
      if (Kind == 1) Image.resize(Gen() % (Image.size() + 1));
      else if (Kind == 2) for (int k = 0; k < 3; ++k) Image[Gen() % 80 % Image.size()] = (unsigned char)Gen();
      FILE* f = std::fopen(Path.c_str(), "wb");
      if (!f) { std::printf("cannot write %s\n", Path.c_str()); return 1; }
      if (!Image.empty()) std::fwrite(Image.data(), 1, Image.size(), f);
      std::fclose(f);
      std::string FromImage, FromFile;
      std::vector<unsigned char> A, B;
      try {
        afec::TWaveFile Wave;
        Wave.OpenForRead(Image.data(), Image.size());
        A.resize(Wave.SampleDataBytes());
        Wave.ReadSampleData(A.data());
      } catch (const afec::TReadableException& e) { FromImage = e.what(); }
      try {
        afec::TWaveFile Wave;
        Wave.OpenForRead(Path);
        unsigned char* Exact = (unsigned char*)std::malloc(Wave.SampleDataBytes() ? Wave.SampleDataBytes() : 1);   // a write past the end is an ASan error
        Wave.ReadSampleData(Exact);
        B.assign(Exact, Exact + Wave.SampleDataBytes());
        std::free(Exact);
        ++FromDisk;
      } catch (const afec::TReadableException& e) { FromFile = e.what(); ++RejectedOnDisk; }
      if (FromImage != FromFile || A != B) ++Differ;
    }
    std::remove(Path.c_str());
    std::printf("wav files: %d read, %d rejected, %d differ from their images\n", FromDisk, RejectedOnDisk, Differ);
    if (Differ) return 1;
  }

  // ---- the oracle's sample-rate conversion (exactly sized input: any read outside [0, n) is an ASan error) ----
  {
    size_t Total = 0;
    for (int Rate : {8000, 11025, 22050, 32000, 44099, 48000, 88200, 96000, 192000, 400000})
      for (int64_t n : {1, 2, 30, 4039, 4040, 4041, 4096, 9001}) {
        float* In = (float*)std::malloc(sizeof(float) * (size_t)n);
        for (int64_t i = 0; i < n; ++i) In[i] = (float)((int)(Gen() % 60001) - 30000);
        int64_t NOut = 0, NWritten = 0;
        float* Out = afx_oracle_resample(In, n, Rate, 44100, &NOut, &NWritten);
        if (NOut < 1 || NWritten > NOut) { std::printf("resample: bad sizes\n"); return 1; }
        Total += (size_t)NOut;
        afx_oracle_free(Out);
        std::free(In);
      }
    std::printf("resample: %zu samples out\n", Total);
  }

  // ---- column encoder ----
  {
    std::vector<double> v(70000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (double)i * 0.5;
    const auto a = afec::ToMsgpack(v.data(), 0), b = afec::ToMsgpack(v.data(), 15), c = afec::ToMsgpack(v.data(), 70000),
               d = afec::ToMsgpack(v.data(), 5000, 14), e = afec::ToMsgpack(v.data(), 0, 14);
    std::printf("msgpack: %zu %zu %zu %zu %zu bytes\n", a.size(), b.size(), c.size(), d.size(), e.size());
    afec::TSampleDescriptors D;
    D.mSpectralCentroid.mValues = {1.0, 2.0};
    D.mRhythmComplexOnsets.mValues.assign(100, 0.25);
    const auto Columns = afec::LowLevelColumns(D, nullptr);
    std::printf("columns: %zu\n", Columns.size());
    // the writer's refill path: one vector reused over files of different lengths, with and without the load info
    std::vector<afec::TColumn> Reused;
    afec::TSampleDataInfo Info = {0.5f, 0.25f, -2205, 90000};
    size_t Bytes = 0;
    for (int Round = 0; Round < 50; ++Round) {
      afec::TSampleDescriptors E;
      const size_t n = (size_t)((Round * 37) % 91);
      E.mSpectralCentroid.mValues.assign(n, 1.5);
      E.mCepstrumBands.mValues.assign((size_t)((Round * 13) % 23), std::array<double, 14>{});
      E.mSpectrumBands.mValues.assign(n / 2, std::array<double, 28>{});
      E.mRhythmPercussiveOnsets.mValues.assign((size_t)(Round * 70000 / 49), 0.125);     // up to the 32-bit array header
      afec::RefillLowLevelColumns(Reused, E, (Round % 3 == 0) ? nullptr : &Info);
      const auto Fresh = afec::LowLevelColumns(E, (Round % 3 == 0) ? nullptr : &Info);
      if (Fresh.size() != Reused.size()) { std::printf("refill: column count differs\n"); return 1; }
      for (size_t i = 0; i < Fresh.size(); ++i) {
        if (Fresh[i].mName != Reused[i].mName || Fresh[i].mType != Reused[i].mType || Fresh[i].mBlob != Reused[i].mBlob ||
            (Fresh[i].mType == afec::TColumn::kReal && Fresh[i].mReal != Reused[i].mReal)) { std::printf("refill: column %zu differs\n", i); return 1; }
        Bytes += Reused[i].mBlob.size();
      }
    }
    std::printf("refill: 50 files, %zu BLOB bytes, identical to fresh columns\n", Bytes);
  }

  // ---- oracle ----
  afx_oracle* o = afx_oracle_create(44100, 2048, 1024);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int64_t n : {0, 1, 511, 512, 640, 2047, 2048, 3071, 3072, 50000, 44100 * 21}) {
    std::vector<double> x((size_t)n);
    for (auto& v : x) v = U(Gen) * std::exp(-(double)(&v - x.data()) / 30000.0);
    for (int Cap : {0, 1}) {
      if (n > 100000 && Cap == 0) continue;
      const int64_t nf = afx_oracle_num_frames(o, n, Cap);
      std::vector<double> rec((size_t)nf * AFXO_RECORD + 1), neigh((size_t)nf * AFXN_RECORD + 1);
      if (afx_oracle_run(o, x.data(), n, Cap, rec.data()) != nf) return 1;
      if (n <= 50000 && afx_oracle_run_neighbours(o, x.data(), n, Cap, neigh.data()) != nf) return 1;
      const int64_t nt = afx_oracle_rhythm_frames(o, n, Cap);
      std::vector<double> onsets((size_t)nt * 2 + 1), sharp((size_t)nt * 2 + 1), odf((size_t)nt * 2 + 1);
      double sc[14];
      if (afx_oracle_run_rhythm(o, x.data(), n, Cap, 44100, n, -100, onsets.data(), sharp.data(), odf.data(), sc) != nt) return 1;
      double eff[3];
      afx_oracle_effective_length(o, x.data(), n, eff);
      double st[13] = {0};
      afx_oracle_calc_statistics(onsets.data(), (int)nt, st);
    }
  }
  {
    // click track: tempo path of the rhythm oracle (autocorrelation, comb filterbank, heuristics)
    std::vector<double> x(44100 * 6, 0.0);
    for (size_t at = 0; at + 2000 < x.size(); at += 11025)
      for (int i = 0; i < 2000; ++i) x[at + (size_t)i] += std::exp(-i / 300.0) * U(Gen);
    const int64_t nt = afx_oracle_rhythm_frames(o, (int64_t)x.size(), 1);
    std::vector<double> onsets((size_t)nt * 2);
    double sc[14];
    afx_oracle_run_rhythm(o, x.data(), (int64_t)x.size(), 1, 44100, (int64_t)x.size(), 0, onsets.data(), nullptr, nullptr, sc);
    std::printf("rhythm oracle: %.0f / %.0f onsets, final tempo %.2f (%.2f)\n", sc[0], sc[6], sc[12], sc[13]);
  }
  {
    std::vector<int16_t> pcm(2 * 30000);
    for (auto& v : pcm) v = (int16_t)(Gen() % 20000) - 10000;
    for (int i = 0; i < 4000; ++i) pcm[(size_t)i] = 0;
    afx_oracle_load_info info;
    double* mono = afx_oracle_load_sample(pcm.data(), 0, 2, 30000, 2048, &info);
    afx_oracle_free(mono);
    int16_t one = 5;
    mono = afx_oracle_load_sample(&one, 0, 1, 1, 2048, &info);
    afx_oracle_free(mono);
  }
  afx_oracle_destroy(o);
  std::printf("sanitize: done\n");
  return 0;
}
