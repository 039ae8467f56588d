// tests/sanitize/fuzz_host_abi.cpp -- TEST INFRASTRUCTURE ONLY.  The C-ABI's host code (afec_amd/csrc/afx_plan.cpp,
// afx_workspace.cpp, afx_batch_plan.cpp, afx_batch_create.cpp, afx_batch_run.cpp, afx_batch_fetch.cpp -- the files the
// round-5 split rewrote) compiled by g++ under AddressSanitizer + UndefinedBehaviorSanitizer (or ThreadSanitizer) on the
// mock device of tests/sanitize/hipstub + mock_kernels.cpp, driven through include/afx.h with fuzzed ragged batches:
//
//   * afx_batch_create: 0 .. 1 100 buffers of 0 .. 300 000 samples (0-frame buffers, exactly one frame, a buffer that
//     claims 2^30 + samples behind the 20 s cap), bad buffers, float / double, every kind of descriptor mask; run, fetch
//     into exactly-sized arrays, statistics, repeated runs (the work-queue counters), destroy;
//   * afx_batch_create_from_raw: every sample type, 1 .. 8 channels, silence at either end, all-silent files, files at other
//     rates (converted; refused: above 16 x the rate, >= 2^30 converted samples), contiguous staging and scattered
//     buffers, >= 768 files (whole-file whitening chunks), the rhythm tracker's long-file path, afx_batch_set_file_info;
//   * device out of memory at the n-th allocation, and a memory limit that only the pool trim gets under;
//   * several threads on one plan.
// The mock kernels assert the chunk-table / queue / placement invariants (mock_kernels.cpp) and write values that depend
// on a frame's own samples only: the driver checks them row by row, so a fetch that unpacks the wrong column or a chunk
// table that sends a frame to the wrong row fails here, without a GPU.
//
// usage: fuzz_host_abi [rounds per thread = 300] [seed = 1] [threads = 1]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "../../include/afx.h"

namespace {

std::atomic<long> g_batches{0}, g_frames{0}, g_refused{0}, g_oom{0};

[[noreturn]] void die(const char* what, long long a = 0, long long b = 0) {
  std::fprintf(stderr, "fuzz_host_abi: FAILED: %s (%lld, %lld); last error: %s\n", what, a, b, afx_last_error());
  std::abort();
}
#define REQUIRE(cond, ...) do { if (!(cond)) die(#cond, ##__VA_ARGS__); } while (0)

struct Rng {
  std::mt19937_64 g;
  explicit Rng(uint64_t seed) : g(seed) {}
  int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)(g() % (uint64_t)(hi - lo + 1)); }
  bool chance(int percent) { return (int)(g() % 100) < percent; }
  template <typename T> T pick(std::initializer_list<T> l) { return *(l.begin() + (size_t)(g() % l.size())); }
};

// the mock frame kernel's value of a frame (mock_kernels.cpp: frame_key)
template <typename T>
double key_of(const T* x, int64_t off) { return (double)x[off] + 0.5 * (double)x[off + 511] + 0.25 * (double)x[off + 1024] + 0.125 * (double)x[off + 2047]; }

int64_t frames_of(int64_t n, int64_t cap) {
  const int64_t len = cap > 0 ? std::min(n, cap) : n;
  return len >= 2048 ? (len - 2048) / 1024 + 1 : 0;
}

uint32_t random_mask(Rng& r, bool allow_whole_buffer) {
  uint32_t m = r.pick<uint32_t>({AFX_D_C2, AFX_D_MFCC | AFX_D_SPECTRAL_RMS | AFX_D_SPECTRAL_CENTROID | AFX_D_SPECTRAL_ROLLOFF, 0xFFu, AFX_D_ALL_LOW_LEVEL,
                                 AFX_D_ALL_PER_FRAME, AFX_D_NEIGHBOURS, AFX_D_MFCC | AFX_D_MAGNITUDE, AFX_D_SPECTRAL_FLUX, AFX_D_BAND_FEATURES | AFX_D_MFCC,
                                 AFX_D_AUTO_CORRELATION | AFX_D_MFCC, AFX_D_F0, AFX_D_SPECTRAL_COMPLEXITY | AFX_D_MFCC, AFX_D_AMPLITUDE_PEAK | AFX_D_AMPLITUDE_RMS,
                                 0u});
  if (m == 0u) m = (uint32_t)r.range(1, 0x3FFFFF) & (AFX_D_ALL_PER_FRAME | AFX_D_MAGNITUDE);
  if (m == 0u) m = AFX_D_MFCC;
  if (r.chance(50)) m |= AFX_D_STATISTICS;
  if (allow_whole_buffer && r.chance(25)) m |= AFX_D_EFFECTIVE_LENGTH;
  if (allow_whole_buffer && r.chance(25)) m |= AFX_D_RHYTHM;
  return m;
}

// the device runs out of memory while a batch is being created: at the n-th allocation from now, or because it is full
// up to (nearly) what is in use -- the idle pooled workspaces included, which ws_reserve gives back before it gives up
void inject_fault(Rng& r) {
  if (r.chance(50)) hipstub::fail_allocation_after(r.range(0, 3));
  else hipstub::set_memory_limit(hipstub::device_bytes_in_use() + (size_t)r.range(0, 1 << 20));
}
void clear_fault() {
  hipstub::fail_allocation_after(-1);
  hipstub::set_memory_limit(0);
}

// ---- afx_batch_create on the caller's buffers ----
void round_create(afx_plan* plan, int64_t cap_samples, Rng& r, bool inject) {
  const bool f64 = r.chance(30);
  const int n_bufs = (int)r.pick<int64_t>({0, 1, 1, 2, 3, 5, 17, 64, 300, 800, 1100});
  const bool giant = cap_samples > 0 && n_bufs > 0 && n_bufs < 20 && r.chance(15);
  const uint32_t mask = random_mask(r, !giant);
  std::vector<std::unique_ptr<float[]>> f32s;
  std::vector<std::unique_ptr<double[]>> f64s;
  std::vector<afx_buf> bufs((size_t)n_bufs);
  std::vector<int64_t> claimed((size_t)n_bufs), frames((size_t)n_bufs);
  std::vector<char> good((size_t)n_bufs, 1);
  const int64_t longest = n_bufs > 300 ? 6000 : (n_bufs > 20 ? 40000 : 300000);
  for (int i = 0; i < n_bufs; ++i) {
    int64_t n = r.pick<int64_t>({0, r.range(1, 2047), 2048, r.range(2049, 4096), r.range(2048, longest), r.range(2048, longest), 3072, 2048 + 1024 * r.range(0, 40)});
    int64_t backed = n;
    if (giant && i == 0) { n = ((int64_t)1 << 30) + r.range(0, 100000); backed = cap_samples + 2048; }   // only the analysed prefix (+ 64) is read
    claimed[(size_t)i] = n;
    void* p = nullptr;
    if (f64) { f64s.emplace_back(new double[(size_t)std::max<int64_t>(backed, 1)]); p = f64s.back().get(); for (int64_t k = 0; k < backed; ++k) f64s.back()[(size_t)k] = (double)(int64_t)(r.g() % 2001) / 1000.0 - 1.0; f32s.emplace_back(nullptr); }
    else { f32s.emplace_back(new float[(size_t)std::max<int64_t>(backed, 1)]); p = f32s.back().get(); for (int64_t k = 0; k < backed; ++k) f32s.back()[(size_t)k] = (float)(int64_t)(r.g() % 2001) / 1000.0f - 1.0f; f64s.emplace_back(nullptr); }
    bufs[(size_t)i] = afx_buf{p, f64 ? AFX_PCM_F64 : AFX_PCM_F32, 0, n};
    if (r.chance(4)) { bufs[(size_t)i].pcm = nullptr; good[(size_t)i] = n == 0; }
    else if (r.chance(3)) { bufs[(size_t)i].n_samples = -5; good[(size_t)i] = 0; }
    else if (r.chance(3)) { bufs[(size_t)i].dtype = 7; good[(size_t)i] = 0; }
    else if (i > 0 && r.chance(3)) { bufs[(size_t)i].dtype = f64 ? AFX_PCM_F32 : AFX_PCM_F64; good[(size_t)i] = 0; bufs[(size_t)i].n_samples = std::min<int64_t>(n, 100); }
    frames[(size_t)i] = good[(size_t)i] ? frames_of(bufs[(size_t)i].n_samples, cap_samples) : 0;
  }
  // (the first VALID buffer decides the batch's PCM type: when buffer 0 is bad and a later one has the other type, that one wins)
  int batch_dtype = -1;
  for (int i = 0; i < n_bufs && batch_dtype < 0; ++i)
    if (bufs[(size_t)i].n_samples >= 0 && (bufs[(size_t)i].n_samples == 0 || bufs[(size_t)i].pcm) && (bufs[(size_t)i].dtype == AFX_PCM_F32 || bufs[(size_t)i].dtype == AFX_PCM_F64)) batch_dtype = bufs[(size_t)i].dtype;
  for (int i = 0; i < n_bufs; ++i) {
    const afx_buf& s = bufs[(size_t)i];
    const bool valid = s.n_samples >= 0 && (s.n_samples == 0 || s.pcm) && (s.dtype == AFX_PCM_F32 || s.dtype == AFX_PCM_F64) && s.dtype == batch_dtype;
    good[(size_t)i] = valid;
    frames[(size_t)i] = valid ? frames_of(s.n_samples, cap_samples) : 0;
  }
  if (inject) inject_fault(r);
  afx_batch* b = nullptr;
  int st = afx_batch_create(plan, n_bufs ? bufs.data() : nullptr, n_bufs, mask, &b);
  clear_fault();
  if (st != AFX_OK) {
    REQUIRE(inject && (st == AFX_ERR_OUT_OF_MEMORY || st == AFX_ERR_HIP), st);
    REQUIRE(b == nullptr);
    ++g_oom;
    return;
  }
  int64_t total = 0;
  for (int64_t f : frames) total += f;
  REQUIRE(afx_batch_total_frames(b) == total, afx_batch_total_frames(b), total);
  afx_batch_info info{};
  REQUIRE(afx_batch_get_info(b, &info) == AFX_OK);
  REQUIRE(total == 0 || (info.chunk_frames >= 1 && info.chunk_frames <= 32 && info.n_chunks >= 1), info.chunk_frames, info.n_chunks);
  if (mask & AFX_D_RHYTHM) {
    std::vector<afx_file_info> fi((size_t)n_bufs);
    for (int i = 0; i < n_bufs; ++i) fi[(size_t)i] = afx_file_info{(int32_t)r.pick<int64_t>({0, 44100, 48000}), (int32_t)r.range(-500, 500), r.range(1, 1 << 20)};
    if (n_bufs > 0 && r.chance(50)) REQUIRE(afx_batch_set_file_info(b, fi.data()) == AFX_OK);
  }
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) { REQUIRE(afx_batch_run(b) == AFX_OK); REQUIRE(afx_batch_sync(b) == AFX_OK); }
    else { float ms = 0; REQUIRE(afx_batch_run_timed(b, (int32_t)r.range(1, 3), &ms) == AFX_OK); }
    // fetch into exactly-sized arrays
    afx_out out{};
    std::vector<double> mfcc, srms, bands, subc, mag, f0, eff;
    std::vector<int64_t> off((size_t)n_bufs + 1, -1);
    std::vector<int32_t> status((size_t)n_bufs, 99);
    if (mask & AFX_D_MFCC) { mfcc.assign((size_t)total * 14, -7.0); out.mfcc = mfcc.data(); }
    if (mask & AFX_D_SPECTRAL_RMS) { srms.assign((size_t)total, -7.0); out.spectral_rms = srms.data(); }
    if (mask & AFX_D_SPECTRUM_BANDS) { bands.assign((size_t)total * 28, -7.0); out.spectrum_bands = bands.data(); }
    if (mask & AFX_D_BAND_FEATURES) { subc.assign((size_t)total * 14, -7.0); out.sub_contrast = subc.data(); }
    if ((mask & AFX_D_MAGNITUDE) && total < 20000) { mag.assign((size_t)total * 1024, -7.0); out.magnitude = mag.data(); }
    if (mask & AFX_D_F0) { f0.assign((size_t)total, -7.0); out.failsafe_f0 = f0.data(); }
    if (mask & AFX_D_EFFECTIVE_LENGTH) { eff.assign((size_t)n_bufs * 3, -7.0); out.effective_length = eff.data(); }
    out.frame_offset = off.data();
    out.buf_status = status.data();
    REQUIRE(afx_batch_fetch(b, &out) == AFX_OK);
    int64_t row = 0;
    for (int i = 0; i < n_bufs; ++i) {
      REQUIRE(off[(size_t)i] == row, i, off[(size_t)i]);
      REQUIRE((status[(size_t)i] == AFX_OK) == (bool)good[(size_t)i], i, status[(size_t)i]);
      for (int64_t f = 0; f < frames[(size_t)i]; ++f, ++row) {
        const double key = f64 ? key_of(f64s[(size_t)i].get(), f * 1024) : key_of(f32s[(size_t)i].get(), f * 1024);
        if (out.mfcc) REQUIRE(mfcc[(size_t)row * 14] == key && mfcc[(size_t)row * 14 + 13] == key + 1e-3 * 13, i, f);
        if (out.spectral_rms && !(mask & (AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX | AFX_D_SPECTRUM_BANDS))) REQUIRE(srms[(size_t)row] == key + 1, i, f);
        if (out.magnitude) REQUIRE(mag[(size_t)row * 1024 + 1023] == std::fabs(key) + 1023, i, f);
        if (out.sub_contrast) REQUIRE(subc[(size_t)row * 14] != -7.0, i, f);
        if (out.failsafe_f0) REQUIRE(f0[(size_t)row] != -7.0, i, f);
      }
    }
    REQUIRE(off[(size_t)n_bufs] == total);
    if (mask & AFX_D_STATISTICS) {
      afx_stats_out so{};
      std::vector<double> smfcc;
      std::vector<int32_t> sst((size_t)n_bufs, 99);
      if (mask & AFX_D_MFCC) { smfcc.assign((size_t)n_bufs * 14 * 13, -7.0); so.mfcc = smfcc.data(); }
      so.stats_status = sst.data();
      REQUIRE(afx_batch_fetch_statistics(b, &so) == AFX_OK);
    }
    if (mask & AFX_D_RHYTHM) {
      std::vector<int64_t> roff((size_t)n_bufs + 1);
      const int64_t rows = afx_batch_rhythm_frames(b, roff.data());
      std::vector<double> onsets((size_t)rows * 2), scalars((size_t)n_bufs * 14), ostats((size_t)n_bufs * 26);
      REQUIRE(afx_batch_fetch_rhythm(b, onsets.data(), scalars.data(), (mask & AFX_D_STATISTICS) ? ostats.data() : nullptr) == AFX_OK);
      std::vector<float> odf((size_t)rows * 2 + 1);
      REQUIRE(afx_batch_fetch_onset_functions(b, odf.data()) == AFX_OK);
    }
  }
  g_frames += total;
  ++g_batches;
  afx_batch_destroy(b);
}

// ---- afx_batch_create_from_raw: decoded files through the LoadSample front end ----
void round_raw(afx_plan* plan, Rng& r, bool inject) {
  const int n = (int)r.pick<int64_t>({1, 2, 7, 64, 200, 513, 900});
  const bool contiguous = r.chance(50);
  const int64_t longest = n > 300 ? 3000 : (n > 20 ? 30000 : 150000);
  struct File { int format, channels, rate; int64_t frames; size_t bytes; bool ok; };
  std::vector<File> files((size_t)n);
  size_t arena_bytes = 0;
  for (File& f : files) {
    f.format = (int)r.pick<int64_t>({AFX_RAW_I16, AFX_RAW_I16, AFX_RAW_I16, AFX_RAW_I24, AFX_RAW_F32, AFX_RAW_I32, AFX_RAW_F64});
    f.channels = (int)r.pick<int64_t>({1, 1, 2, 2, 3, 8});
    f.rate = (int)r.pick<int64_t>({0, 44100, 44100, 44100, 48000, 22050, 96000, 11025, 8000});
    f.frames = r.pick<int64_t>({1, r.range(2, 500), r.range(500, longest), r.range(500, longest), 44100, 2048});
    f.ok = true;
    const int bps = f.format == AFX_RAW_I16 ? 2 : f.format == AFX_RAW_I24 ? 3 : f.format == AFX_RAW_F64 ? 8 : 4;
    f.bytes = (size_t)f.frames * f.channels * bps;
    arena_bytes += (f.bytes + 15) & ~(size_t)15;
  }
  // headers that lie: a rate above 16 x the analyser's, and one that makes the conversion 2^30 samples or more
  if (r.chance(30)) { files[0].rate = 800000; files[0].ok = false; }
  if (n > 1 && r.chance(30)) { files[1].rate = 1; files[1].frames = std::max<int64_t>(files[1].frames, 30000); files[1].ok = false;
    const int bps = files[1].format == AFX_RAW_I16 ? 2 : files[1].format == AFX_RAW_I24 ? 3 : files[1].format == AFX_RAW_F64 ? 8 : 4;
    arena_bytes -= (files[1].bytes + 15) & ~(size_t)15; files[1].bytes = (size_t)files[1].frames * files[1].channels * bps; arena_bytes += (files[1].bytes + 15) & ~(size_t)15; }
  std::unique_ptr<unsigned char[]> arena(new unsigned char[arena_bytes + 16]);
  std::vector<std::unique_ptr<unsigned char[]>> scattered;
  std::vector<afx_raw> raws((size_t)n);
  unsigned char* base = arena.get() + ((16 - ((uintptr_t)arena.get() & 15)) & 15);
  size_t at = 0;
  for (int i = 0; i < n; ++i) {
    File& f = files[(size_t)i];
    unsigned char* p;
    if (contiguous) { p = base + at; at += (f.bytes + 15) & ~(size_t)15; }
    else { scattered.emplace_back(new unsigned char[f.bytes ? f.bytes : 1]); p = scattered.back().get(); }
    const int bps = (int)(f.bytes / (size_t)(f.frames * f.channels));
    const int64_t lead = r.chance(40) ? r.range(0, f.frames) : 0, trail = r.chance(40) ? r.range(0, f.frames - lead) : 0;
    for (int64_t k = 0; k < f.frames; ++k)
      for (int c = 0; c < f.channels; ++c) {
        unsigned char* q = p + ((size_t)k * f.channels + c) * bps;
        const bool silent = k < lead || k >= f.frames - trail;
        const double v = silent ? 0.0 : (double)(int64_t)(r.g() % 20001) / 10000.0 - 1.0;
        switch (f.format) {
          case AFX_RAW_I16: { const int16_t s = (int16_t)(v * 32767.0); std::memcpy(q, &s, 2); break; }
          case AFX_RAW_I24: { const int32_t s = (int32_t)(v * 8388607.0); q[0] = (unsigned char)s; q[1] = (unsigned char)(s >> 8); q[2] = (unsigned char)(s >> 16); break; }
          case AFX_RAW_I32: { const int32_t s = (int32_t)(v * 2147483000.0); std::memcpy(q, &s, 4); break; }
          case AFX_RAW_F64: std::memcpy(q, &v, 8); break;
          default: { const float s = (float)v; std::memcpy(q, &s, 4); }
        }
      }
    raws[(size_t)i] = afx_raw{p, f.format, f.channels, f.rate, 0, f.frames};
    if (r.chance(2)) { raws[(size_t)i].channels = 9; f.ok = false; }
    else if (r.chance(2)) { raws[(size_t)i].format = 11; f.ok = false; }
  }
  if (!contiguous || true) {
    // (a bad file in a contiguous staging buffer takes no place in the device arena: the library then uploads file by file)
  }
  const uint32_t mask = random_mask(r, true);
  std::vector<afx_load_info> info((size_t)n);
  if (inject) inject_fault(r);
  afx_batch* b = nullptr;
  const int st = afx_batch_create_from_raw(plan, raws.data(), n, mask, &b, r.chance(80) ? info.data() : nullptr);
  clear_fault();
  if (st != AFX_OK) {
    REQUIRE(inject && (st == AFX_ERR_OUT_OF_MEMORY || st == AFX_ERR_HIP), st);
    ++g_oom;
    return;
  }
  REQUIRE(afx_batch_run(b) == AFX_OK);
  int32_t stride = 0, offsets[AFX_NUM_SERIES], widths[AFX_NUM_SERIES];
  REQUIRE(afx_batch_record_layout(b, &stride, offsets, widths) == AFX_OK);
  const int64_t total = afx_batch_total_frames(b);
  std::vector<double> rec((size_t)total * stride), stats((mask & AFX_D_STATISTICS) ? (size_t)n * stride * 13 : 0), eff((mask & AFX_D_EFFECTIVE_LENGTH) ? (size_t)n * 3 : 0);
  std::vector<int64_t> off((size_t)n + 1);
  std::vector<int32_t> status((size_t)n);
  REQUIRE(afx_batch_fetch_records(b, rec.data(), stats.empty() ? nullptr : stats.data(), off.data(), status.data(), eff.empty() ? nullptr : eff.data()) == AFX_OK);
  for (int i = 0; i < n; ++i) {
    const File& f = files[(size_t)i];
    if (!f.ok) { REQUIRE(status[(size_t)i] != AFX_OK, i); REQUIRE(off[(size_t)i + 1] == off[(size_t)i], i); ++g_refused; continue; }
    REQUIRE(status[(size_t)i] == AFX_OK, i, status[(size_t)i]);
    REQUIRE(off[(size_t)i + 1] - off[(size_t)i] >= 1, i);         // LoadSample pads every file to at least one frame
  }
  if (mask & AFX_D_RHYTHM) {
    std::vector<int64_t> roff((size_t)n + 1);
    const int64_t rows = afx_batch_rhythm_frames(b, roff.data());
    std::vector<double> onsets((size_t)rows * 2), scalars((size_t)n * 14), ostats((size_t)n * 26);
    REQUIRE(afx_batch_fetch_rhythm(b, onsets.data(), scalars.data(), (mask & AFX_D_STATISTICS) ? ostats.data() : nullptr) == AFX_OK);
  }
  {
    const int pick = (int)r.range(0, n - 1);
    std::vector<double> samples(4096);
    REQUIRE(afx_batch_fetch_samples(b, pick, samples.data(), 4096) == AFX_OK);
  }
  REQUIRE(afx_batch_run(b) == AFX_OK);       // once more: queue counters, pooled buffers
  REQUIRE(afx_batch_sync(b) == AFX_OK);
  g_frames += total;
  ++g_batches;
  afx_batch_destroy(b);
}

void worker(afx_plan* capped, afx_plan* uncapped, int rounds, uint64_t seed, bool inject) {
  Rng r(seed);
  for (int k = 0; k < rounds; ++k) {
    const bool oom = inject && r.chance(20);
    switch (k % 3) {
      case 0: round_create(capped, 882000, r, oom); break;
      case 1: round_raw(capped, r, oom); break;
      default: round_create(uncapped, 0, r, oom);
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 300;
  const uint64_t seed = argc > 2 ? (uint64_t)std::atoll(argv[2]) : 1;
  const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
  afx_plan_desc desc = {44100, 2048, 1024, 0, AFX_PRECISION_F64, 20000, AFX_FRAME_KERNEL_AUTO, 0};
  afx_plan *capped = nullptr, *uncapped = nullptr, *pinned = nullptr;
  REQUIRE(afx_plan_create(&desc, &capped) == AFX_OK);
  desc.max_analysis_ms = 0;
  desc.frame_kernel = AFX_FRAME_KERNEL_HALFWAVE;      // every batch the half-wave layout serves takes it: its chunk pairs and queue
  REQUIRE(afx_plan_create(&desc, &uncapped) == AFX_OK);
  desc.device = 3;
  REQUIRE(afx_plan_create(&desc, &pinned) == AFX_ERR_NO_DEVICE && pinned == nullptr);   // one mock device: ordinal 3 does not exist
  REQUIRE(afx_device_count() == 1);
  REQUIRE(afx_plan_probe_device(capped) == AFX_OK);
  afx_plan_set_blocking_wait(capped, 1);             // the sleeping waits' path (polls an event)

  if (threads <= 1) worker(capped, uncapped, rounds, seed, true);
  else {
    // several threads on the two plans (allocation faults are process-wide in the stub: off here)
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker, capped, uncapped, rounds, seed * 1000 + (uint64_t)t, false);
    for (std::thread& t : pool) t.join();
  }

  // a device that is full until the idle pool gives its memory back: two idle workspaces, a large one and a small one (the
  // next batch is handed the small one, must grow it, finds the device full, ws_reserve trims the pool -- the large one
  // goes -- and the second attempt succeeds)
  {
    std::vector<float> x(2048 + 1024 * 4000, 0.25f);
    afx_buf big{x.data(), AFX_PCM_F32, 0, (int64_t)x.size()}, tiny{x.data(), AFX_PCM_F32, 0, 4096}, mid{x.data(), AFX_PCM_F32, 0, 2048 + 1024 * 3000};
    afx_batch *a = nullptr, *t = nullptr, *b = nullptr;
    REQUIRE(afx_batch_create(uncapped, &big, 1, AFX_D_ALL_LOW_LEVEL, &a) == AFX_OK);
    REQUIRE(afx_batch_create(uncapped, &tiny, 1, AFX_D_ALL_LOW_LEVEL, &t) == AFX_OK);
    afx_batch_destroy(a);
    afx_batch_destroy(t);        // the pool hands out the workspace released last: the tiny one
    hipstub::set_memory_limit(hipstub::device_bytes_in_use() + 4096);
    REQUIRE(afx_batch_create(uncapped, &mid, 1, AFX_D_ALL_LOW_LEVEL, &b) == AFX_OK);
    REQUIRE(afx_batch_run(b) == AFX_OK);
    afx_batch_destroy(b);
    hipstub::set_memory_limit(0);
  }
  // a lost device: every call fails with AFX_ERR_HIP, the probe says so, nothing crashes; afterwards it is back
  {
    hipstub::lose_device(0, true);
    REQUIRE(afx_plan_probe_device(capped) == AFX_ERR_HIP);
    std::vector<float> x(4096, 0.5f);
    afx_buf one{x.data(), AFX_PCM_F32, 0, 4096};
    afx_batch* b = nullptr;
    REQUIRE(afx_batch_create(capped, &one, 1, AFX_D_C2, &b) != AFX_OK && b == nullptr);
    hipstub::lose_device(0, false);
    REQUIRE(afx_plan_probe_device(capped) == AFX_OK);
    REQUIRE(afx_batch_create(capped, &one, 1, AFX_D_C2, &b) == AFX_OK);
    afx_batch_destroy(b);
  }
  afx_plan_destroy(capped);
  afx_plan_destroy(uncapped);
  REQUIRE(hipstub::device_bytes_in_use() == 0, (long long)hipstub::device_bytes_in_use());
  REQUIRE(hipstub::live_streams() == 0 && hipstub::live_events() == 0, hipstub::live_streams(), hipstub::live_events());
  std::printf("fuzz_host_abi: %ld batches, %ld frames, %ld refused files, %ld injected allocation failures survived; %d thread(s), seed %llu: clean\n",
              g_batches.load(), g_frames.load(), g_refused.load(), g_oom.load(), threads, (unsigned long long)seed);
  return 0;
}
