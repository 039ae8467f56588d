// tests/sanitize/mock_kernels.cpp -- TEST INFRASTRUCTURE ONLY: the launchers of afx_internal.h for the mock device of
// tests/sanitize/hipstub.  No descriptor is computed here.  Each "kernel" walks exactly the tables its HIP counterpart
// walks, touches every byte it would read or write ("device memory" is the host heap, so AddressSanitizer sees an
// allocation the planner sized too small or an offset that points past an arena), asserts the invariants the real
// kernels rely on without checking them, and leaves values that depend on nothing but the frame's own samples -- so
// the host layers above (the C-ABI's fetches, afec::TCrawler, the row digests) can be tested for "the same content
// gives the same row whatever batch / device it was analysed on" without a GPU.
//
// Invariants asserted (afx_batch_plan.cpp is what establishes them):
//   * chunk tables: 1 <= nframes <= 32; the rows [frame0, frame0 + nframes) of all chunks tile [0, total) exactly once;
//     a chunk with kChunkFirstOfBuffer starts a buffer; with the autocorrelation selected a chunk that is not a buffer's
//     last has an even number of frames; `remaining` covers the chunk's frames;
//   * work queues (QueueBook): the counter in device memory equals the base the host passes, at every launch;
//   * whitening chunk table: chunk_first is a prefix sum over buffers, a buffer's chunks tile its frames in order, a chunk
//     that does not start a buffer exists only with need_follow (and then the follower checkpoints are allocated);
//   * statistics regimes: small_rows / need_long describe the frame offsets that were uploaded;
//   * rhythm tracker: file rows are prefix sums, every 512/128 frame lies inside the arena, long-file tables are
//     consistent with the files' long_slot;
//   * LoadSample / resample tables: every file's source and destination ranges lie inside their arenas (touched).
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../afec_amd/csrc/afx_internal.h"
#include "../../include/afx.h"

namespace afx {

namespace {

[[noreturn]] void broken(const char* what, long long a = 0, long long b = 0) {
  std::fprintf(stderr, "mock kernel: invariant violated: %s (%lld, %lld)\n", what, a, b);
  std::abort();
}
#define MOCK_CHECK(cond, ...) do { if (!(cond)) broken(#cond, ##__VA_ARGS__); } while (0)

inline double pcm_at(const void* pcm, int dtype, int64_t idx, double scale) {
  if (dtype == kPcmF64) return ((const volatile double*)pcm)[idx];
  const double v = (double)((const volatile float*)pcm)[idx];
  return dtype == kPcmScaledF32 ? v * scale : v;
}
// what a frame "is" for the mock: a few of its own samples (first, two inside, last: the frame's whole extent is read)
inline double frame_key(const void* pcm, int dtype, int64_t off, double scale) {
  return pcm_at(pcm, dtype, off, scale) + 0.5 * pcm_at(pcm, dtype, off + 511, scale) + 0.25 * pcm_at(pcm, dtype, off + 1024, scale) +
         0.125 * pcm_at(pcm, dtype, off + kFft - 1, scale);
}
inline void put(double* rec, const RecordLayout& lay, int64_t row, int32_t off, int width, double v) {
  if (off < 0) return;
  for (int k = 0; k < width; ++k) rec[row * lay.stride + off + k] = v + 1e-3 * k;
}
void check_queue(const WorkQueue& q, unsigned items) {
  if (!q.counter) return;
  MOCK_CHECK(*q.counter == q.base, (long long)*q.counter, (long long)q.base);
  *q.counter += items;
}

// the rows of a chunk table tile [0, total) exactly once (total = one past the largest row)
void check_tiling(const Chunk* chunks, int n_chunks) {
  int64_t top = 0, sum = 0;
  for (int c = 0; c < n_chunks; ++c) {
    MOCK_CHECK(chunks[c].nframes >= 1 && chunks[c].nframes <= 32767, c, chunks[c].nframes);
    MOCK_CHECK(chunks[c].frame0 >= 0, c, chunks[c].frame0);
    top = std::max<int64_t>(top, (int64_t)chunks[c].frame0 + chunks[c].nframes);
    sum += chunks[c].nframes;
  }
  MOCK_CHECK(sum == top, sum, top);
  std::vector<unsigned char> seen((size_t)top, 0);
  for (int c = 0; c < n_chunks; ++c)
    for (int f = 0; f < chunks[c].nframes; ++f) {
      MOCK_CHECK(!seen[(size_t)chunks[c].frame0 + f], c, f);
      seen[(size_t)chunks[c].frame0 + f] = 1;
    }
}

void frames_body(const FrameArgs& a, int pcm_dtype, bool halfwave) {
  MOCK_CHECK(a.chunks != nullptr && a.pcm != nullptr);
  check_tiling(a.chunks, a.n_chunks);
  for (int c = 0; c < a.n_chunks; ++c) {
    const Chunk& ch = a.chunks[c];
    MOCK_CHECK(ch.nframes <= 32, c, ch.nframes);    // the frame kernels' chunk: a wave's run of at most 32 frames
    MOCK_CHECK((ch.sample_off & 3) == 0 || !(ch.flags & kChunkFirstOfBuffer), c, ch.sample_off);   // buffers start 16-byte aligned (float)
    for (int f = 0; f < ch.nframes; ++f) {
      const int64_t row = (int64_t)ch.frame0 + f;
      const double key = frame_key(a.pcm, pcm_dtype, ch.sample_off + (int64_t)f * kHop, ch.scale);
      if (a.rec) {
        const RecordLayout& l = a.lay;
        if (a.mask & AFX_D_MFCC) put(a.rec, l, row, l.mfcc, 14, key);
        if (a.mask & AFX_D_SPECTRAL_RMS) put(a.rec, l, row, l.srms, 1, key + 1);
        if (a.mask & AFX_D_SPECTRAL_CENTROID) put(a.rec, l, row, l.centroid, 1, key + 2);
        if (a.mask & AFX_D_SPECTRAL_SPREAD) put(a.rec, l, row, l.spread, 1, key + 3);
        if (a.mask & AFX_D_SPECTRAL_SKEWNESS) put(a.rec, l, row, l.skew, 1, key + 4);
        if (a.mask & AFX_D_SPECTRAL_KURTOSIS) put(a.rec, l, row, l.kurt, 1, key + 5);
        if (a.mask & AFX_D_SPECTRAL_ROLLOFF) put(a.rec, l, row, l.rolloff, 1, key + 6);
        if (a.mask & AFX_D_SPECTRAL_FLATNESS) put(a.rec, l, row, l.flatness, 1, key + 7);
        if (a.mask & AFX_D_SPECTRUM_BANDS) put(a.rec, l, row, l.bands, 28, key + 8);
        if (a.mask & AFX_D_AMPLITUDE_PEAK) put(a.rec, l, row, l.amp_peak, 1, std::fabs(key));
        if (a.mask & AFX_D_AMPLITUDE_RMS) put(a.rec, l, row, l.amp_rms, 1, std::fabs(key) * 0.5);
      }
      if (a.mag_out)
        for (int k = 0; k < kHalf; ++k) a.mag_out[row * kHalf + k] = std::fabs(key) + k;
      if (halfwave && a.stat_tmp)
        for (int k = 0; k < kStatTmp; ++k) a.stat_tmp[row * kStatTmp + k] = key + k;
    }
  }
  if (halfwave && a.mag_out)    // the half-wave full classes send the stores of frames past a chunk's end to a spare row
    for (int k = 0; k < kHalf; ++k) a.mag_out[a.mag_spare_row * kHalf + k] = 0.0;
}

}  // namespace

// ---- the host-side rules that live in the .hip files (restated; tests/test_sanitize_cpu.py compares them with the
// product library's own functions over every mask) ----
int frames_feature_class(uint32_t mask) {
  if (mask == 1u) return 0;
  if (mask & ((1u << 8) | (1u << 9) | (1u << 10) | (1u << 13))) return 2;
  return 1;
}
int frames_waves_per_block(uint32_t) { return 8; }
int frames32_waves_per_block() { return 8; }
int frames32_stat_tmp_doubles() { return kStatTmp; }
bool frames_use_halfwave(uint32_t mask, int precision, int pcm_dtype) {
  return (mask & 1u) && !(mask & ~(0x3FFFu | kFramesWholeSpectrum | kFramesStatsLater)) && precision == 0 &&
         (pcm_dtype == kPcmF32 || pcm_dtype == kPcmScaledF32);
}
int frames32_class(uint32_t mask) {
  if (mask == 1u) return 0;
  if (!(mask & ~0xFFu)) return 1;
  if (mask & kFramesStatsLater) return 4;
  return (mask & kFramesWholeSpectrum) ? 3 : 2;
}
int load_scan_blocks_per_file(int n_files) { return n_files >= 1024 ? 1 : (n_files >= 64 ? 8 : 64); }
int64_t resample_blocks(int64_t n_out) { return (n_out + 4096 - 1) / 4096; }

hipError_t launch_frames(const FrameArgs& a, int precision, int pcm_dtype, int grid_blocks, hipStream_t) {
  if (a.n_chunks <= 0) return hipSuccess;
  MOCK_CHECK(precision == 0 && grid_blocks >= 1);
  MOCK_CHECK(a.win && a.t1 && a.t2 && a.post && a.melw && a.dct);
  MOCK_CHECK(!((a.mask & AFX_D_MAGNITUDE) && !a.mag_out));
  frames_body(a, pcm_dtype, false);
  return hipSuccess;
}

hipError_t launch_frames32(const FrameArgs& a, int grid_blocks, hipStream_t, int64_t total_frames, bool scaled) {
  if (a.n_chunks <= 0) return hipSuccess;
  MOCK_CHECK(grid_blocks >= 1 && a.win32 && a.tw32 && a.post32 && a.melw32 && a.dct && a.queue);
  MOCK_CHECK(*a.queue == a.queue_base, (long long)*a.queue, (long long)a.queue_base);
  *a.queue += (unsigned)((a.n_chunks + 1) / 2);
  const int cls = frames32_class(a.mask);
  MOCK_CHECK(cls == 0 || a.stat_tmp != nullptr, cls);
  MOCK_CHECK(cls < 2 || a.mag_out != nullptr, cls);
  MOCK_CHECK(a.mag_spare_row == total_frames, a.mag_spare_row, total_frames);
  int64_t rows = 0;
  for (int c = 0; c < a.n_chunks; ++c) rows += a.chunks[c].nframes;
  MOCK_CHECK(rows == total_frames, rows, total_frames);
  frames_body(a, scaled ? kPcmScaledF32 : kPcmF32, true);
  return hipSuccess;
}

hipError_t launch_stats32_finish(const FrameArgs& a, hipStream_t, int64_t total_frames) {
  MOCK_CHECK(a.stat_tmp != nullptr && a.rec != nullptr);
  const RecordLayout& l = a.lay;
  for (int64_t row = 0; row < total_frames; ++row) {
    const double key = a.stat_tmp[row * kStatTmp] + a.stat_tmp[row * kStatTmp + kStatTmp - 1];
    if (a.mask & AFX_D_SPECTRAL_RMS) put(a.rec, l, row, l.srms, 1, key + 1);
    if (a.mask & AFX_D_SPECTRAL_CENTROID) put(a.rec, l, row, l.centroid, 1, key + 2);
    if (a.mask & AFX_D_SPECTRAL_SPREAD) put(a.rec, l, row, l.spread, 1, key + 3);
    if (a.mask & AFX_D_SPECTRAL_SKEWNESS) put(a.rec, l, row, l.skew, 1, key + 4);
    if (a.mask & AFX_D_SPECTRAL_KURTOSIS) put(a.rec, l, row, l.kurt, 1, key + 5);
    if (a.mask & AFX_D_SPECTRAL_ROLLOFF) put(a.rec, l, row, l.rolloff, 1, key + 6);
    if (a.mask & AFX_D_SPECTRAL_FLATNESS) put(a.rec, l, row, l.flatness, 1, key + 7);
  }
  return hipSuccess;
}

hipError_t launch_bands(const BandArgs& a, hipStream_t) {
  if (a.n_chunks <= 0) return hipSuccess;
  MOCK_CHECK(a.mag != nullptr && a.chunks != nullptr && a.rec != nullptr);
  MOCK_CHECK(!(a.flags & kBandsStats) || a.stat_tmp != nullptr);
  check_queue(a.queue, (unsigned)a.n_chunks);
  check_tiling(a.chunks, a.n_chunks);
  const RecordLayout& l = a.lay;
  for (int c = 0; c < a.n_chunks; ++c) {
    const Chunk& ch = a.chunks[c];
    for (int f = 0; f < ch.nframes; ++f) {
      const int64_t row = (int64_t)ch.frame0 + f;
      // flux and the sub-band flux compare a frame with its predecessor in the same buffer; a buffer's first with itself
      const bool first = (ch.flags & kChunkFirstOfBuffer) && f == 0;
      MOCK_CHECK(first || row > 0, row);
      const int64_t prev = first ? row : row - 1;
      const double key = a.mag[row * kHalf] + a.mag[row * kHalf + kHalf - 1] + 0.5 * (a.mag[prev * kHalf] + a.mag[prev * kHalf + kHalf - 1]);
      if (a.flags & kBandsFeatures) {
        put(a.rec, l, row, l.sub_rms, 14, key);
        put(a.rec, l, row, l.sub_flat, 14, key + 1);
        put(a.rec, l, row, l.sub_flux, 14, key + 2);
        put(a.rec, l, row, l.sub_cplx, 14, key + 3);
        put(a.rec, l, row, l.sub_contrast, 14, key + 4);
        put(a.rec, l, row, l.contrast, 1, key + 5);
      }
      if (a.flags & kBandsFlux) put(a.rec, l, row, l.flux, 1, key + 6);
      if (a.flags & kBandsSpectrum) put(a.rec, l, row, l.bands, 28, key + 8);
      if (a.flags & kBandsStats)
        for (int k = 0; k < kStatTmp; ++k) a.stat_tmp[row * kStatTmp + k] = key + k;
    }
  }
  return hipSuccess;
}

namespace {
// the chunk walk of the time-domain kernels: `remaining` = samples of the buffer from the chunk's first frame on
void time_body(const TimeArgs& a, bool acorr, void (*emit)(const TimeArgs&, int64_t, double)) {
  MOCK_CHECK(a.pcm && a.chunks && a.remaining && a.rec);
  check_tiling(a.chunks, a.n_chunks);
  for (int c = 0; c < a.n_chunks; ++c) {
    const Chunk& ch = a.chunks[c];
    const int64_t rem = a.remaining[c];
    const int64_t need = (int64_t)(ch.nframes - 1) * kHop + kFft;
    MOCK_CHECK(rem >= need, rem, need);
    const bool more_frames_follow = rem < (1 << 30) && rem - (int64_t)ch.nframes * kHop >= kFft;
    if (acorr) MOCK_CHECK(!more_frames_follow || (ch.nframes % 2) == 0, c, ch.nframes);   // a frame's partner does not depend on the cut
    for (int f = 0; f < ch.nframes; ++f) {
      const int64_t off = ch.sample_off + (int64_t)f * kHop;
      double key = frame_key(a.pcm, a.pcm_dtype, off, ch.scale);
      if (acorr) {   // the second rising-slope search may look up to 33 samples past the frame when the buffer has them
        const int64_t left = rem - (int64_t)f * kHop;
        const int64_t reach = std::min<int64_t>(left, kFft + 33);
        key += 1e-6 * pcm_at(a.pcm, a.pcm_dtype, off + reach - 1, ch.scale);
      }
      emit(a, (int64_t)ch.frame0 + f, key);
    }
  }
}
}  // namespace

hipError_t launch_hop(const TimeArgs& a, hipStream_t) {
  if (a.n_chunks <= 0) return hipSuccess;
  time_body(a, false, [](const TimeArgs& t, int64_t row, double key) {
    put(t.rec, t.lay, row, t.lay.silence, 1, key < 0 ? 1.0 : 0.0);
    put(t.rec, t.lay, row, t.lay.envelope, 1, std::fabs(key));
    if (t.amplitude & AFX_D_AMPLITUDE_PEAK) put(t.rec, t.lay, row, t.lay.amp_peak, 1, std::fabs(key));
    if (t.amplitude & AFX_D_AMPLITUDE_RMS) put(t.rec, t.lay, row, t.lay.amp_rms, 1, std::fabs(key) * 0.5);
  });
  return hipSuccess;
}
hipError_t launch_pitch(const TimeArgs& a, hipStream_t) {
  if (a.n_chunks <= 0) return hipSuccess;
  MOCK_CHECK(a.t1 && a.t2 && a.post);
  check_queue(a.queue, (unsigned)a.n_chunks);
  time_body(a, false, [](const TimeArgs& t, int64_t row, double key) {
    put(t.rec, t.lay, row, t.lay.f0, 1, 100.0 + key);
    put(t.rec, t.lay, row, t.lay.f0_conf, 1, 0.5);
    if (t.hop_here) {
      put(t.rec, t.lay, row, t.lay.silence, 1, key < 0 ? 1.0 : 0.0);
      put(t.rec, t.lay, row, t.lay.envelope, 1, std::fabs(key));
      if (t.amplitude & AFX_D_AMPLITUDE_PEAK) put(t.rec, t.lay, row, t.lay.amp_peak, 1, std::fabs(key));
      if (t.amplitude & AFX_D_AMPLITUDE_RMS) put(t.rec, t.lay, row, t.lay.amp_rms, 1, std::fabs(key) * 0.5);
    }
  });
  return hipSuccess;
}
hipError_t launch_acorr(const TimeArgs& a, hipStream_t) {
  if (a.n_chunks <= 0) return hipSuccess;
  MOCK_CHECK(a.t1 && a.t2 && a.post);
  check_queue(a.queue, (unsigned)a.n_chunks);
  time_body(a, true, [](const TimeArgs& t, int64_t row, double key) { put(t.rec, t.lay, row, t.lay.autocorr, 1, key); });
  return hipSuccess;
}

hipError_t launch_whiten(const WhitenArgs& a, hipStream_t) {
  if (a.n_bufs <= 0) return hipSuccess;
  MOCK_CHECK(a.mag && a.frame_offset && a.chunk_first && a.chunks && a.rec);
  check_queue(a.queue, (unsigned)a.n_chunks);
  MOCK_CHECK(a.chunk_first[0] == 0 && a.chunk_first[a.n_bufs] == a.n_chunks, a.chunk_first[a.n_bufs], a.n_chunks);
  bool several = false;
  for (int i = 0; i < a.n_bufs; ++i) {
    const int c0 = a.chunk_first[i], c1 = a.chunk_first[i + 1];
    MOCK_CHECK(c0 <= c1, i);
    int64_t row = a.frame_offset[i];
    for (int c = c0; c < c1; ++c) {
      const Chunk& ch = a.chunks[c];
      MOCK_CHECK(ch.frame0 == row, c, row);                                    // a buffer's chunks tile its frames in order
      MOCK_CHECK(((ch.flags & kChunkFirstOfBuffer) != 0) == (c == c0), c, i);
      MOCK_CHECK(ch.nframes >= 1, c);
      if (c > c0) several = true;
      row += ch.nframes;
    }
    MOCK_CHECK(row == a.frame_offset[i + 1], i, row);
  }
  MOCK_CHECK(!several || a.need_follow, a.n_chunks);                            // start states of inner chunks come from follow_kernel
  if (a.need_follow && (a.mask & AFX_D_SPECTRAL_COMPLEXITY)) {
    MOCK_CHECK(a.follower != nullptr);
    for (int c = 0; c < a.n_chunks; ++c)
      for (int k = 0; k < kHalf; ++k) a.follower[(size_t)c * kHalf + k] = 0.0;
  }
  const RecordLayout& l = a.lay;
  for (int c = 0; c < a.n_chunks; ++c)
    for (int f = 0; f < a.chunks[c].nframes; ++f) {
      const int64_t row = (int64_t)a.chunks[c].frame0 + f;
      const double key = a.mag[row * kHalf + 1] + a.mag[row * kHalf + kHalf - 1];
      if (a.mask & AFX_D_SPECTRAL_COMPLEXITY) put(a.rec, l, row, l.complexity, 1, std::floor(key));
      if (a.mask & AFX_D_F0) put(a.rec, l, row, l.f0_safe, 1, key + (l.f0 >= 0 ? a.rec[row * l.stride + l.f0] : 0.0));
      if (a.mask & AFX_D_SPECTRAL_INHARMONICITY) put(a.rec, l, row, l.inharm, 1, 0.0);
      if (a.mask & AFX_D_TRISTIMULUS) { put(a.rec, l, row, l.tri1, 1, 0.0); put(a.rec, l, row, l.tri2, 1, 0.0); put(a.rec, l, row, l.tri3, 1, 0.0); }
    }
  return hipSuccess;
}

hipError_t launch_effective_length(const void* pcm, int pcm_dtype, const BufSpan* spans, int n_bufs, double floor48, double floor24,
                                   double floor12, int32_t* out, hipStream_t) {
  MOCK_CHECK(n_bufs <= 0 || (spans && out));
  MOCK_CHECK(floor48 < floor24 && floor24 < floor12);
  for (int i = 0; i < n_bufs; ++i) {
    const BufSpan& s = spans[i];
    for (int j = 0; j < 3; ++j) { out[i * 6 + 2 * j] = INT_MAX; out[i * 6 + 2 * j + 1] = -1; }
    if (s.n <= 0) continue;
    MOCK_CHECK(pcm != nullptr);
    (void)pcm_at(pcm, pcm_dtype, s.off, s.scale);
    (void)pcm_at(pcm, pcm_dtype, s.off + s.n - 1, s.scale);
    for (int j = 0; j < 3; ++j) { out[i * 6 + 2 * j] = 0; out[i * 6 + 2 * j + 1] = (int32_t)std::min<int64_t>(s.n - 1, INT_MAX - 1); }
  }
  return hipSuccess;
}

hipError_t launch_stats(const StatsArgs& a, hipStream_t) {
  if (a.n_bufs <= 0 || a.stride <= 0) return hipSuccess;
  MOCK_CHECK(a.frame_offset && a.stats);
  int small = 0, need_long = 0;
  for (int i = 0; i < a.n_bufs; ++i) {
    const int64_t n = a.frame_offset[i + 1] - a.frame_offset[i];
    MOCK_CHECK(n >= 0, i, n);
    if (n >= 2 && n <= 128) small = std::max(small, (int)n); else need_long = 1;
  }
  MOCK_CHECK(small == a.small_rows && need_long == a.need_long, a.small_rows, a.need_long);
  MOCK_CHECK(a.frame_offset[a.n_bufs] == 0 || a.rec != nullptr);
  for (int i = 0; i < a.n_bufs; ++i) {
    const int64_t r0 = a.frame_offset[i], n = a.frame_offset[i + 1] - r0;
    for (int col = 0; col < a.stride; ++col) {
      double lo = 0, hi = 0, sum = 0;
      for (int64_t r = 0; r < n; ++r) {
        const double v = a.rec[(r0 + r) * a.stride + col];
        if (r == 0 || v < lo) lo = v;
        if (r == 0 || v > hi) hi = v;
        sum += v;
      }
      double* s = a.stats + ((size_t)i * a.stride + col) * 13;
      for (int k = 0; k < 13; ++k) s[k] = 0.0;
      s[AFX_S_MIN] = lo; s[AFX_S_MAX] = hi; s[AFX_S_MEAN] = n ? sum / (double)n : 0.0;
    }
  }
  return hipSuccess;
}

hipError_t launch_rhythm(const RhythmArgs& a, hipStream_t) {
  if (a.n_files <= 0) return hipSuccess;
  MOCK_CHECK(a.files && a.scalars && a.window && a.tw256 && a.ut512 && a.canny && a.rayleigh);
  MOCK_CHECK(a.lds_frames <= kRhythmLdsFrames && a.medspan >= 3 && a.medspan <= 256);
  int64_t rows = 0;
  int n_long = 0;
  for (int i = 0; i < a.n_files; ++i) {
    const RhythmFile& f = a.files[i];
    MOCK_CHECK(f.frame0 == rows && f.frames >= 0, i, f.frame0);
    MOCK_CHECK(std::isfinite(f.duration_s) && std::isfinite(f.offset_s), i);
    if (f.frames > 0) {
      MOCK_CHECK(a.pcm && a.odf && a.onsets && a.scratch);
      const double first = pcm_at(a.pcm, a.pcm_dtype, f.sample_off, f.scale);
      const double last = pcm_at(a.pcm, a.pcm_dtype, f.sample_off + (int64_t)(f.frames - 1) * 128 + 511, f.scale);
      for (int64_t t = 0; t < f.frames; ++t) {
        const double v = pcm_at(a.pcm, a.pcm_dtype, f.sample_off + t * 128, f.scale);
        a.odf[(rows + t) * 2] = (float)v; a.odf[(rows + t) * 2 + 1] = (float)-v;
        a.onsets[(rows + t) * 2] = (t % 16 == 0) ? std::fabs(v) : 0.0; a.onsets[(rows + t) * 2 + 1] = 0.0;
      }
      for (int k = 0; k < AFX_NUM_RHYTHM_SCALARS; ++k) a.scalars[(size_t)i * AFX_NUM_RHYTHM_SCALARS + k] = first + last + k + f.duration_s + f.offset_s;
    } else {
      for (int k = 0; k < AFX_NUM_RHYTHM_SCALARS; ++k) a.scalars[(size_t)i * AFX_NUM_RHYTHM_SCALARS + k] = 0.0;
    }
    if (f.long_slot > 0) {
      ++n_long;
      MOCK_CHECK(f.long_slot <= a.n_long && a.long_files && a.long_files[f.long_slot - 1] == i, i, f.long_slot);
      MOCK_CHECK(f.frames >= kRhythmLongFrames, i, f.frames);
    }
    rows += f.frames;
  }
  MOCK_CHECK(rows == a.total_frames, rows, a.total_frames);
  MOCK_CHECK(n_long == a.n_long, n_long, a.n_long);
  if (a.total_frames > 0)
    for (int k = 0; k < 8; ++k) { a.scratch[(size_t)k * a.total_frames] = 0.0; a.scratch[(size_t)(k + 1) * a.total_frames - 1] = 0.0; }
  if (a.n_long > 0) {
    MOCK_CHECK(a.long_round_off && a.long_frame_off && a.long_polar && a.long_den);
    MOCK_CHECK(a.n_files <= kRhythmLongBatchFiles, a.n_files);
    MOCK_CHECK(a.long_round_off[0] == 0 && a.long_frame_off[0] == 0 && a.long_round_off[a.n_long] == a.long_rounds);
    for (int s = 0; s < a.n_long; ++s) {
      const RhythmFile& f = a.files[a.long_files[s]];
      MOCK_CHECK(a.long_round_off[s + 1] - a.long_round_off[s] == (f.frames + 15) / 16, s);
      const int64_t padded = (f.frames + kRhythmLongPad - 1) / kRhythmLongPad * kRhythmLongPad;
      MOCK_CHECK(a.long_frame_off[s + 1] - a.long_frame_off[s] == padded, s);
    }
    const int64_t lrows = a.long_frame_off[a.n_long];
    // (magnitude, phase) pairs, then the follower's float per bin behind them
    MOCK_CHECK((const void*)a.long_den == (const void*)(a.long_polar + lrows * 256), lrows);
    a.long_polar[0] = float2{0, 0}; a.long_polar[lrows * 256 - 1] = float2{0, 0};
    a.long_den[0] = 0.f; a.long_den[lrows * 256 - 1] = 0.f;
  }
  return hipSuccess;
}

namespace {
inline int raw_bps(int format) { return format == AFX_RAW_I16 ? 2 : format == AFX_RAW_I24 ? 3 : format == AFX_RAW_F64 ? 8 : 4; }
// first channel of sample frame k of a file, in the "16-bit float" range (the mock's mono mix)
inline float raw_sample(const unsigned char* raw, const LoadFile& f, int64_t k) {
  const unsigned char* p = raw + f.raw_off + (size_t)k * f.channels * raw_bps(f.format);
  switch (f.format) {
    case AFX_RAW_I16: { int16_t v; std::memcpy(&v, p, 2); return (float)v; }
    case AFX_RAW_I24: { const int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8; return (float)v / 256.0f; }
    case AFX_RAW_I32: { int32_t v; std::memcpy(&v, p, 4); return (float)v / 65536.0f; }
    case AFX_RAW_F64: { double v; std::memcpy(&v, p, 8); return (float)(v * 32768.0); }
    case kRawMonoFloat: { float v; std::memcpy(&v, p, 4); return v; }
    default: { float v; std::memcpy(&v, p, 4); return v * 32768.0f; }
  }
}
}  // namespace

hipError_t launch_load_scan(const unsigned char* raw, const LoadFile* files, int n_files, double silence_floor, void* partial_scratch,
                            LoadScan* scan, hipStream_t) {
  if (n_files <= 0) return hipSuccess;
  MOCK_CHECK(raw && files && partial_scratch && scan && silence_floor > 0);
  std::memset(partial_scratch, 0, (size_t)n_files * load_scan_blocks_per_file(n_files) * 16);
  for (int i = 0; i < n_files; ++i) {
    const LoadFile& f = files[i];
    LoadScan s{};
    s.amplification = 1.0; s.lead = INT_MAX; s.trail = -1;
    if (f.n_frames > 0) {
      MOCK_CHECK(f.channels >= 1 && f.channels <= 8, i, f.channels);
      // the whole file is read (every channel of the last frame too)
      (void)*(const volatile unsigned char*)(raw + f.raw_off + (size_t)f.n_frames * f.channels * raw_bps(f.format) - 1);
      float peak = 0.f;
      for (int64_t k = 0; k < f.n_frames; ++k) {
        const float v = std::fabs(raw_sample(raw, f, k));
        peak = std::max(peak, v);
        s.sum_sq += (double)(v / 32768.0f) * (double)(v / 32768.0f);
      }
      s.max_amp = peak;
      s.amplification = peak > 0.f ? 32768.0 / (double)peak : 1.0;
      for (int64_t k = 0; k < f.n_frames; ++k)
        if ((double)std::fabs(raw_sample(raw, f, k)) * s.amplification > silence_floor) { if (s.lead == INT_MAX) s.lead = (int32_t)k; s.trail = (int32_t)k; }
    }
    scan[i] = s;
  }
  return hipSuccess;
}

hipError_t launch_load_write(const unsigned char* raw, const LoadFile* files, const LoadPlace* place, int n_files, float* arena, hipStream_t) {
  if (n_files <= 0) return hipSuccess;
  MOCK_CHECK(raw && files && place);
  for (int i = 0; i < n_files; ++i) {
    const LoadPlace& p = place[i];
    if (p.out_n <= 0) continue;
    MOCK_CHECK(arena != nullptr);
    MOCK_CHECK(p.start_pad >= 0 && p.audible >= 0 && p.lead >= 0 && p.lead + p.audible <= files[i].n_frames, i, p.lead + p.audible);
    MOCK_CHECK((p.out_off & 3) == 0, i, p.out_off);
    // the slot is written whole (pads and slack too: the pooled arena is not cleared between batches)
    const int64_t slot = (p.out_n + 3) & ~(int64_t)3;
    for (int64_t k = 0; k < slot; ++k) {
      const int64_t src = k - p.start_pad;
      arena[p.out_off + k] = (k < p.out_n && src >= 0 && src < p.audible) ? raw_sample(raw, files[i], p.lead + src) : 0.0f;
    }
  }
  return hipSuccess;
}

hipError_t launch_resample(unsigned char* raw, const ResampleFile* files, int n_files, int64_t n_blocks, int64_t max_n_in,
                           ResampleGroup* groups, const float* filter, hipStream_t) {
  if (n_files <= 0) return hipSuccess;
  MOCK_CHECK(raw && files && groups && filter);
  int64_t blocks = 0, slots = 0, longest = 0;
  for (int i = 0; i < n_files; ++i) {
    const ResampleFile& f = files[i];
    MOCK_CHECK(f.n_in > 0 && f.n_out > 0 && f.factor >= 1.0 / 16.0, i);
    MOCK_CHECK(f.block_off == blocks && f.group_off == slots, i, f.block_off);
    MOCK_CHECK((f.mono_off & 15) == 0 && (f.out_off & 15) == 0, i);
    const LoadFile src{f.raw_off, f.n_in, f.channels, f.format};
    float* mono = (float*)(raw + f.mono_off);
    for (int64_t k = 0; k < f.n_in + 2 * kResampleMargin; ++k) {
      const int64_t s = k - kResampleMargin;
      mono[k] = (s >= 0 && s < f.n_in) ? raw_sample(raw, src, s) : 0.0f;
    }
    float* out = (float*)(raw + f.out_off);
    for (int64_t k = 0; k < f.n_out; ++k) {
      const int64_t s = std::min<int64_t>(f.n_in - 1, (int64_t)((double)k / f.factor));
      out[k] = mono[kResampleMargin + s];
    }
    for (int64_t g = 0; g < (f.n_out + 15) / 16; ++g) groups[slots + g] = ResampleGroup{};
    blocks += resample_blocks(f.n_out);
    slots += (f.n_out + 15) / 16;
    longest = std::max(longest, f.n_in);
  }
  MOCK_CHECK(blocks == n_blocks && longest == max_n_in, blocks, n_blocks);
  return hipSuccess;
}

}  // namespace afx
