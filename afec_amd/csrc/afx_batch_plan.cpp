// afec_amd/csrc/afx_batch_plan.cpp -- what a batch launches and on what: the record layout of a descriptor mask, where
// every buffer's analysed prefix lies in the PCM arena, which layout of the STFT kernel serves the batch, how the frame
// loop (SampleAnalyser.cpp:760-764, 814) is cut into per-wave chunks, the whitening kernels' own chunk table, the rhythm
// tracker's file table, and the device buffers all of that needs.  See afx_host.h for the map of the host side.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "afx_host.h"

namespace afx {
namespace host {

#define AFX_FIELD(name, lay, width, bit) {&afx_out::name, &afx_stats_out::name, &afx::RecordLayout::lay, width, bit}
const FieldDesc kFields[AFX_NUM_SERIES] = {
    AFX_FIELD(mfcc, mfcc, 14, AFX_D_MFCC),
    AFX_FIELD(spectral_rms, srms, 1, AFX_D_SPECTRAL_RMS),
    AFX_FIELD(spectral_centroid, centroid, 1, AFX_D_SPECTRAL_CENTROID),
    AFX_FIELD(spectral_spread, spread, 1, AFX_D_SPECTRAL_SPREAD),
    AFX_FIELD(spectral_skewness, skew, 1, AFX_D_SPECTRAL_SKEWNESS),
    AFX_FIELD(spectral_kurtosis, kurt, 1, AFX_D_SPECTRAL_KURTOSIS),
    AFX_FIELD(spectral_rolloff, rolloff, 1, AFX_D_SPECTRAL_ROLLOFF),
    AFX_FIELD(spectral_flatness, flatness, 1, AFX_D_SPECTRAL_FLATNESS),
    AFX_FIELD(spectral_flux, flux, 1, AFX_D_SPECTRAL_FLUX),
    AFX_FIELD(spectrum_bands, bands, 28, AFX_D_SPECTRUM_BANDS),
    AFX_FIELD(amplitude_peak, amp_peak, 1, AFX_D_AMPLITUDE_PEAK),
    AFX_FIELD(amplitude_rms, amp_rms, 1, AFX_D_AMPLITUDE_RMS),
    AFX_FIELD(sub_rms, sub_rms, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_flatness, sub_flat, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_flux, sub_flux, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_complexity, sub_cplx, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_contrast, sub_contrast, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(spectral_contrast, contrast, 1, AFX_D_BAND_FEATURES),
    AFX_FIELD(amplitude_silence, silence, 1, AFX_D_AMPLITUDE_SILENCE),
    AFX_FIELD(amplitude_envelope, envelope, 1, AFX_D_AMPLITUDE_ENVELOPE),
    AFX_FIELD(spectral_complexity, complexity, 1, AFX_D_SPECTRAL_COMPLEXITY),
    AFX_FIELD(auto_correlation, autocorr, 1, AFX_D_AUTO_CORRELATION),
    AFX_FIELD(f0, f0, 1, AFX_D_F0),
    AFX_FIELD(f0_confidence, f0_conf, 1, AFX_D_F0),
    AFX_FIELD(failsafe_f0, f0_safe, 1, AFX_D_F0),
    AFX_FIELD(spectral_inharmonicity, inharm, 1, AFX_D_SPECTRAL_INHARMONICITY),
    AFX_FIELD(tristimulus1, tri1, 1, AFX_D_TRISTIMULUS),
    AFX_FIELD(tristimulus2, tri2, 1, AFX_D_TRISTIMULUS),
    AFX_FIELD(tristimulus3, tri3, 1, AFX_D_TRISTIMULUS),
};
#undef AFX_FIELD

afx::RecordLayout make_layout(uint32_t mask) {
  afx::RecordLayout l{};
  int off = 0;
  for (const FieldDesc& f : kFields) {
    l.*(f.off) = -1;
    if (mask & f.bit) { l.*(f.off) = off; off += f.width; }
  }
  l.stride = off;
  return l;
}

// the frame kernel sees the spectral bits; storing the magnitudes is its AFX_D_MAGNITUDE path
uint32_t frames_mask(uint32_t mask) {
  uint32_t m = mask & kSpectralBits;
  if (mask & kNeedsMagnitudes) m |= AFX_D_MAGNITUDE;
  // who reads bins above 768 of the stored magnitudes: the caller (AFX_D_MAGNITUDE) and the whitening kernels
  if (mask & (AFX_D_MAGNITUDE | AFX_D_SPECTRAL_COMPLEXITY | AFX_D_F0)) m |= afx::kFramesWholeSpectrum;
  // bands_kernel runs for this mask anyway and the spectral statistics are wanted: on the half-wave layout it takes them
  // from the stored magnitudes (seven sums more over rows it holds), the frame kernel only stores (magnitude class)
  if ((mask & 0xFEu) && (mask & (AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX | AFX_D_SPECTRUM_BANDS))) m |= afx::kFramesStatsLater;
  return m;
}

bool mask_ok(uint32_t mask) {
  return (mask & ~(uint32_t)AFX_D_STATISTICS) != 0 &&
         !(mask & ~(uint32_t)(AFX_D_ALL_PER_FRAME | AFX_D_MAGNITUDE | AFX_D_STATISTICS | AFX_D_EFFECTIVE_LENGTH | AFX_D_RHYTHM));
}

// SampleDurationInSeconds / OnsetOffsetInSeconds of every buffer (SampleAnalyser.cpp:1001-1004): TAudioMath::SamplesToMs
// is float arithmetic (AudioMath.inl:134-137) and the division by the int 1000 stays in float
void set_rhythm_context(afx_batch* b, const afx_file_info* info) {
  for (int32_t i = 0; i < b->n_bufs; ++i) {
    const int own_rate = (i < (int32_t)b->file_rate.size() && b->file_rate[(size_t)i] > 0) ? b->file_rate[(size_t)i] : b->plan->desc.sample_rate;
    const int rate = (info && info[i].original_sample_rate > 0) ? info[i].original_sample_rate : own_rate;
    const int samples = (int)(info ? info[i].original_samples : b->file_samples[(size_t)i]);
    const int offset = info ? info[i].data_offset : b->file_offset[(size_t)i];
    afx::RhythmFile& rf = b->rt_files[(size_t)i];
    rf.duration_s = (double)(((float)samples / ((float)rate / 1000.0f)) / 1000);
    rf.offset_s = (double)(((float)offset / ((float)rate / 1000.0f)) / 1000);
  }
  b->rt_files_dirty = true;
}

namespace {

// What the steps of build_batch share.
struct Planner {
  afx_plan* plan;
  afx_batch* b;
  const BatchSource& src;
  uint32_t mask;        // descriptor bits only (AFX_D_STATISTICS taken out)
  bool want_stats;
  uint32_t fmask;       // frames_mask(mask)
  size_t esz;           // bytes per arena sample
  int64_t arena = 0;    // samples of the whole arena
  int64_t frames = 0;
  int waves_per_block = 0;
  int64_t slots = 0;    // wave (half-wave) slots of the chip for the frame kernel
  int64_t length(int i) const { return (*src.lengths)[(size_t)i]; }
  // kPcmScaledF32 (the LoadSample front end): the arena holds the float mono signal, the buffer's FinalScaling rides
  // along in the chunk table / buffer spans / rhythm file records
  double scale_of(int i) const { return src.scales ? (*src.scales)[(size_t)i] : 1.0; }
};

// arena offsets: every buffer starts on a 16-byte boundary and only the analysed prefix
// (SampleAnalyser.cpp:760-764) of buffers that yield at least one frame is kept
void place_buffers(Planner& p) {
  afx_batch* b = p.b;
  const afx_plan* plan = p.plan;
  const int n_bufs = p.src.n_bufs;
  b->arena_off.assign((size_t)n_bufs, 0);
  b->used.assign((size_t)n_bufs, 0);
  int64_t arena = 0, frames = 0;
  for (int i = 0; i < n_bufs; ++i) {
    b->frame_offset[i] = frames;
    if (b->buf_status[i] != AFX_OK) continue;
    const int64_t f = num_frames(plan, p.length(i));
    int64_t keep = 0;
    if (f > 0) {
      keep = (f - 1) * plan->desc.hop_size + plan->desc.fft_size;
      // the second rising-slope search of CalcAutoCorrelation (SampleAnalyser.cpp:2343-2356) may look up to
      // 33 samples past the last frame when the buffer has them
      if (p.mask & AFX_D_AUTO_CORRELATION) keep = std::min<int64_t>(p.length(i), keep + 64);
      frames += f;
    }
    // the rhythm tracker's 512/128 frames reach up to the end of the analysed prefix (SampleAnalyser.cpp:991)
    if (p.mask & AFX_D_RHYTHM) keep = std::max(keep, analysed_length(plan, p.length(i)));
    // CalcEffectiveLength scans the whole buffer, also beyond the analysed 20 s (SampleAnalyser.cpp:754)
    if (p.mask & AFX_D_EFFECTIVE_LENGTH) keep = p.length(i);
    if (keep > 0) {
      b->used[i] = keep;
      b->arena_off[i] = arena;
      arena += (keep + 3) & ~(int64_t)3;
    }
  }
  b->frame_offset[n_bufs] = frames;
  b->total_frames = frames;
  p.arena = arena;
  p.frames = frames;
}

// Which layout of the STFT kernel serves the batch.  The half-wave kernel pays a longer prologue per chunk: it serves
// batches that give every half-wave slot several frames; smaller ones (one short file per call) stay with the 64-lane
// kernel -- unless the plan pins the choice (afx_plan_desc.frame_kernel: a caller whose results must not depend on how
// its files were batched, the crawler, pins it; the two layouts round differently).
void choose_frame_kernel(Planner& p) {
  afx_batch* b = p.b;
  const afx_plan* plan = p.plan;
  b->halfwave = plan->halfwave && frames_use_halfwave(p.fmask, plan->desc.precision, p.src.dtype) &&
                (plan->halfwave == 2 || p.frames >= 8 * (int64_t)plan->cu_count * frames32_waves_per_block() * 2);
  // the half-wave full class always leaves the magnitudes (bands_kernel takes flux, the 28 bands and the sub-band
  // descriptors from them)
  if (b->halfwave && frames32_class(p.fmask) >= 2) b->mag_wanted = true;
  p.waves_per_block = b->halfwave ? frames32_waves_per_block() : frames_waves_per_block(p.fmask);
  p.slots = (int64_t)plan->cu_count * p.waves_per_block * (b->halfwave ? 2 : 1);
}

// Chunking: K consecutive frames per wave; enough chunks to fill the chip, long enough to amortise the 2048-sample
// lead-in of each chunk.  K minimises a cost model: long chunks for big batches, one round of short chunks when the
// batch barely fills the chip.
int choose_chunk_frames(const Planner& p) {
  const afx_batch* b = p.b;
  // what a chunk costs before its first frame, in frames: the 64-lane kernel loads one hop; the half-wave kernel loads
  // the overlap rows and the window table and starts its hop DMA (measured on the C4 share, profiles/r04)
  constexpr double kHalfwavePrologue = 0.5;
  int K = 32;
  double best = 1e300;
  // (acorr_kernel transforms frames two at a time, frames 2 i and 2 i + 1 of a buffer as the real and imaginary part of
  // one complex sequence: chunks of an even number of frames keep a frame's partner -- and with it the rounding of
  // its result -- independent of how the batch was cut)
  const int k_step = (p.mask & AFX_D_AUTO_CORRELATION) ? 2 : 1;
  for (int k = k_step; k <= 32; k += k_step) {
    int64_t nchunks = 0;
    for (int i = 0; i < p.src.n_bufs; ++i) nchunks += (b->frame_offset[i + 1] - b->frame_offset[i] + k - 1) / k;
    const double prologue = b->halfwave ? kHalfwavePrologue : 0.5;
    // The 64-lane frame kernel gives every wave the same number of chunks: rounds of the wave slots x (frames per chunk
    // + lead-in).  Every kernel of the half-wave batches draws its chunks from a work queue: the slots share the work
    // (frames + lead-ins) evenly and run dry within a fraction of a chunk of each other.
    // (the 0.4: measured -- C3, 1 000 files of 82 frames: K = 6, 39.3 M frames/s against 35.8 M with the K = 22 of the
    // rounds model; the headline batch K = 32, 499 against 495 M with K = 25; the C4 share K = 10, 41.6 against 42.3 M
    // with K = 8; profiles/r04)
    constexpr double kTailWeight = 0.4;
    const double cost = b->halfwave ? ((double)p.frames + (double)nchunks * prologue) / (double)p.slots + kTailWeight * k
                                    : (double)((nchunks + p.slots - 1) / p.slots) * (k + prologue);
    if (cost < best - 1e-9 || (std::fabs(cost - best) <= 1e-9 && k > K)) { best = cost; K = k; }
  }
  return K;
}

// the chunk table every per-frame kernel walks (+ the samples that remain behind a chunk's start, for the time-domain kernels)
void cut_chunks(Planner& p, int K) {
  afx_batch* b = p.b;
  const afx_plan* plan = p.plan;
  std::vector<Chunk>& chunks = b->h_chunks;
  std::vector<ChunkRemaining>& remaining = b->h_remaining;
  b->chunk_frames = K;
  // When the half-wave frame kernel is the only consumer of the chunk table (it draws chunks from a work queue),
  // the last part of every buffer is cut into short chunks and the table is ordered long chunks first: the waves
  // run dry within a quarter of a long chunk's time of each other instead of a whole one.
  const bool guided = b->halfwave && K >= 8 && !(p.mask & (kTimeBits | kWhitenBits | AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX));
  const int Ks = guided ? K / 4 : K;
  for (int i = 0; i < p.src.n_bufs; ++i) {
    const int64_t f = b->frame_offset[i + 1] - b->frame_offset[i];
    const int64_t f_long = guided ? (f * 7 / 8) / K * K : f;   // frames covered by chunks of K
    for (int64_t f0 = 0; f0 < f;) {
      const int k = (f0 < f_long) ? K : Ks;
      remaining.push_back((ChunkRemaining)std::min<int64_t>(p.length(i) - f0 * plan->desc.hop_size, 1 << 30));
      Chunk c;
      c.sample_off = b->arena_off[i] + f0 * plan->desc.hop_size;
      c.frame0 = (int32_t)(b->frame_offset[i] + f0);
      c.nframes = (int16_t)std::min<int64_t>(k, f - f0);
      c.flags = (int16_t)(f0 == 0 ? kChunkFirstOfBuffer : 0);
      c.scale = p.scale_of(i);
      chunks.push_back(c);
      f0 += k;
    }
  }
  if (guided)
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& x, const Chunk& y) { return x.nframes > y.nframes; });
  b->n_chunks = (int)chunks.size();
  // one workgroup per CU (its LDS holds the shared tables plus one exchange plane per wave);
  // waves walk the chunk list with a grid stride
  const int64_t wave_items = b->halfwave ? (b->n_chunks + 1) / 2 : b->n_chunks;
  b->grid_blocks = (int)std::min<int64_t>((wave_items + p.waves_per_block - 1) / p.waves_per_block, (int64_t)plan->cu_count);
  if (b->grid_blocks < 1) b->grid_blocks = 1;
}

// The whitening kernels walk their own chunk table.  Their only state across frames is the follower, which starts
// from the reset value at a buffer's first frame: in a batch of thousands of short files a chunk is the whole file --
// every chunk then starts from the reset state and follow_kernel (one more pass over the 8 KiB of magnitudes per frame)
// is not needed; long files, and batches too small to fill the chip with one wave per file, keep chunks of K frames
// whose start states follow_kernel provides.  (Same values either way: the follower is a max / multiply recurrence that
// both paths run frame by frame in the same order -- tests/test_gpu_load.py compares them bit for bit.)
void cut_whitening_chunks(Planner& p) {
  afx_batch* b = p.b;
  const int n_bufs = p.src.n_bufs, K = b->chunk_frames;
  constexpr int64_t kWholeFileFrames = 128;
  const bool whole_files = n_bufs >= 768;     // one wave per file on at least three quarters of the chip's 1 024 SIMDs
  std::vector<Chunk>& wchunks = b->h_wchunks;
  std::vector<int32_t>& chunk_first = b->h_chunk_first;
  chunk_first.assign((size_t)n_bufs + 1, 0);
  for (int i = 0; i < n_bufs; ++i) {
    const int64_t f = b->frame_offset[i + 1] - b->frame_offset[i];
    chunk_first[(size_t)i] = (int32_t)wchunks.size();
    const int64_t step = (whole_files && f <= kWholeFileFrames) ? std::max<int64_t>(f, 1) : K;
    if (f > step) b->need_follow = true;
    for (int64_t f0 = 0; f0 < f; f0 += step) {
      Chunk c;
      c.sample_off = b->arena_off[i] + f0 * p.plan->desc.hop_size;
      c.frame0 = (int32_t)(b->frame_offset[i] + f0);
      c.nframes = (int16_t)std::min<int64_t>(step, f - f0);
      c.flags = (int16_t)(f0 == 0 ? kChunkFirstOfBuffer : 0);
      c.scale = p.scale_of(i);
      wchunks.push_back(c);
    }
  }
  chunk_first[(size_t)n_bufs] = (int32_t)wchunks.size();
  b->n_wchunks = (int)wchunks.size();
}

// Device memory of a batch.  Buffers the kernels only write or read (records, magnitudes, statistics ...) are reserved
// one by one; the small tables the HOST fills (chunk tables, offsets, the rhythm tracker's file records: nine of them for
// a crawler's batch) are collected and travel as ONE block -- one page-locked source, one hipMemcpyAsync, one device
// buffer the batch's pointers point into.  Until round 5 each was its own copy from pageable memory: every copy is
// staged by the runtime on the calling thread and completes through the runtime's event thread (~35 us of host CPU
// each; that thread was 0.75 of the 2.2 CPUs a crawl kept busy, tools/thread_cpu.py).
struct Reserver {
  afx_plan* plan;
  Workspace* ws;
  hipStream_t stream;
  hipError_t e = hipSuccess;
  const char* what = "";
  struct Table { void** dst; size_t off, bytes; };
  std::vector<Table> tables;
  std::vector<unsigned char> image;   // the tables back to back, each on a 64-byte boundary
  // a host-filled table: into the block (the pointer is set by flush)
  template <typename T>
  bool table(T** dst, const void* src, size_t bytes, const char* name) {
    what = name;
    *dst = nullptr;
    if (!bytes) return true;
    const size_t off = (image.size() + 63) & ~(size_t)63;
    image.resize(off + bytes);
    std::memcpy(image.data() + off, src, bytes);
    tables.push_back(Table{(void**)dst, off, bytes});
    return true;
  }
  // device memory the kernels fill or read
  template <typename T>
  bool operator()(Workspace::Buf& buf, T** dst, size_t bytes, const char* name) {
    what = name;
    if ((e = ws_reserve(plan, buf, bytes)) != hipSuccess) return false;
    *dst = (T*)buf.p;
    return true;
  }
  // the collected tables: one upload, then the batch's pointers
  bool flush() {
    what = "tables";
    if (tables.empty()) return true;
    if ((e = ws_reserve(plan, ws->tables, image.size())) != hipSuccess) return false;
    if ((e = ws_pin_reserve(ws, image.size())) != hipSuccess) return false;
    std::memcpy(ws->h_pin, image.data(), image.size());
    if ((e = hipMemcpyAsync(ws->tables.p, ws->h_pin, image.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return false;
    for (const Table& t : tables) *t.dst = (unsigned char*)ws->tables.p + t.off;
    return true;
  }
};

// device tables of the per-frame kernels: chunk tables, work-queue counters, records, magnitudes, statistics
int reserve_frame_buffers(Planner& p, Reserver& r) {
  afx_batch* b = p.b;
  Workspace& w = *b->ws;
  const int n_bufs = p.src.n_bufs;
  const int64_t frames = p.frames;
  auto failed = [&]() { return hip_fail(r.e, r.what); };
  if (b->n_chunks > 0) {
    if (!r.table(&b->d_chunks, b->h_chunks.data(), b->h_chunks.size() * sizeof(Chunk), "chunks")) return failed();
    if (!w.queues.attached()) {
      // zeroed once: every launch advances its counter by its number of items (afx_host.h: QueueBook)
      unsigned* d = nullptr;
      if (!r(w.queue, &d, kQueueSlots * sizeof(unsigned), "queue")) return failed();
      if ((r.e = hipMemsetAsync(d, 0, kQueueSlots * sizeof(unsigned), b->stream)) != hipSuccess) return failed();
      w.queues.attach(d);
    }
    if (b->halfwave && p.fmask != 1u)   // statistics class: raw sums per frame for its closed-form kernel
      if (!r(w.stat_tmp, &b->d_stat_tmp, (size_t)frames * frames32_stat_tmp_doubles() * sizeof(double), "stat_tmp")) return failed();
    if (p.mask & kTimeBits)
      if (!r.table(&b->d_rem, b->h_remaining.data(), b->h_remaining.size() * sizeof(ChunkRemaining), "remaining")) return failed();
    if (p.mask & kWhitenBits) {
      cut_whitening_chunks(p);
      if (!r.table(&b->d_wchunks, b->h_wchunks.data(), b->h_wchunks.size() * sizeof(Chunk), "whitening chunks")) return failed();
      if (!r.table(&b->d_chunk_first, b->h_chunk_first.data(), b->h_chunk_first.size() * sizeof(int32_t), "chunk_first")) return failed();
      if ((p.mask & AFX_D_SPECTRAL_COMPLEXITY) && b->need_follow)
        if (!r(w.follower, &b->d_follower, (size_t)b->n_wchunks * kHalf * sizeof(double), "follower")) return failed();
    }
  }
  if (n_bufs > 0 && (p.mask & AFX_D_EFFECTIVE_LENGTH)) {
    std::vector<BufSpan>& spans = b->h_spans;
    spans.resize((size_t)n_bufs);
    for (int i = 0; i < n_bufs; ++i) spans[(size_t)i] = BufSpan{b->arena_off[i], b->used[i], p.scale_of(i)};
    if (!r.table(&b->d_spans, spans.data(), spans.size() * sizeof(BufSpan), "spans")) return failed();
    if (!r(w.efflen, &b->d_efflen, (size_t)n_bufs * 6 * sizeof(int32_t), "efflen")) return failed();
  }
  if (frames > 0 && b->lay.stride > 0)
    if (!r(w.rec, &b->d_rec, (size_t)frames * b->lay.stride * sizeof(double), "rec")) return failed();
  if (frames > 0 && b->mag_wanted)   // (+ 1 row: the half-wave full class sends the stores of frames past a chunk's end there)
    if (!r(w.mag, &b->d_mag, (size_t)(frames + 1) * kHalf * sizeof(double), "mag")) return failed();
  if ((p.want_stats || (p.mask & kWhitenBits)) && n_bufs > 0 && b->lay.stride > 0)
    if (!r.table(&b->d_frame_offset, b->frame_offset.data(), b->frame_offset.size() * sizeof(int64_t), "frame_offset")) return failed();
  if (p.want_stats && n_bufs > 0 && b->lay.stride > 0)
    if (!r(w.stats, &b->d_stats, (size_t)n_bufs * b->lay.stride * 13 * sizeof(double), "stats")) return failed();
  return AFX_OK;
}

// The rhythm tracker's file table -- 512/128 frames of every buffer's analysed prefix: for (n = 0; n + 511 < length;
// n += 128), SampleAnalyser.cpp:991 -- and, for a batch of few files, which of them take the three-kernel path: such a
// batch does not fill the chip with one workgroup per file, and a long file is hundreds of dependent rounds for its
// workgroup (a lone 20 s file: 430 rounds, 5.7 ms).  Returns the blob of the long files' tables (empty: none).
std::vector<unsigned char> plan_rhythm_files(Planner& p, int64_t* long_rows) {
  afx_batch* b = p.b;
  const afx_plan* plan = p.plan;
  const int n_bufs = p.src.n_bufs;
  b->rt_offset.assign((size_t)n_bufs + 1, 0);
  b->rt_files.assign((size_t)n_bufs, RhythmFile{});
  b->file_samples.assign((size_t)n_bufs, 0);
  b->file_offset.assign((size_t)n_bufs, 0);
  if (p.src.file_rate) b->file_rate = *p.src.file_rate;
  int64_t rows = 0;
  for (int i = 0; i < n_bufs; ++i) {
    b->rt_offset[(size_t)i] = rows;
    RhythmFile& rf = b->rt_files[(size_t)i];
    rf.sample_off = b->arena_off[i];
    rf.scale = p.scale_of(i);
    rf.frame0 = rows;
    const int64_t len = (b->buf_status[i] == AFX_OK) ? std::min(analysed_length(plan, p.length(i)), b->used[i]) : 0;
    rf.frames = (len >= 512) ? (int32_t)((len - 512) / 128 + 1) : 0;
    rows += rf.frames;
    b->file_samples[(size_t)i] = p.src.file_samples ? (*p.src.file_samples)[(size_t)i] : p.length(i);
    b->file_offset[(size_t)i] = p.src.file_offset ? (*p.src.file_offset)[(size_t)i] : 0;
  }
  b->rt_offset[(size_t)n_bufs] = rows;
  std::vector<unsigned char> blob;
  *long_rows = 0;
  if (n_bufs > kRhythmLongBatchFiles) return blob;
  std::vector<int32_t> long_files, long_round_off;
  std::vector<int64_t> long_frame_off;
  int32_t rounds = 0;
  int64_t lrows = 0;
  for (int i = 0; i < n_bufs; ++i)
    if (b->rt_files[(size_t)i].frames >= kRhythmLongFrames) {
      long_files.push_back(i);
      long_round_off.push_back(rounds);
      long_frame_off.push_back(lrows);
      b->rt_files[(size_t)i].long_slot = (int32_t)long_files.size();
      rounds += (b->rt_files[(size_t)i].frames + 15) / 16;
      lrows += (b->rt_files[(size_t)i].frames + kRhythmLongPad - 1) / kRhythmLongPad * kRhythmLongPad;
    }
  long_round_off.push_back(rounds);
  long_frame_off.push_back(lrows);
  b->rt_n_long = (int32_t)long_files.size();
  b->rt_long_rounds = rounds;
  if (b->rt_n_long > 0) {
    const size_t nl = long_files.size();
    // layout: int64 frame offsets [nl + 1], then int32 round offsets [nl + 1], then int32 file indices [nl]
    blob.resize((nl + 1) * 8 + (nl + 1) * 4 + nl * 4);
    std::memcpy(blob.data(), long_frame_off.data(), (nl + 1) * 8);
    std::memcpy(blob.data() + (nl + 1) * 8, long_round_off.data(), (nl + 1) * 4);
    std::memcpy(blob.data() + (nl + 1) * 12, long_files.data(), nl * 4);
    *long_rows = lrows;
  }
  return blob;
}

int reserve_rhythm_buffers(Planner& p, Reserver& r) {
  afx_batch* b = p.b;
  Workspace& w = *b->ws;
  const int n_bufs = p.src.n_bufs;
  auto failed = [&]() { return hip_fail(r.e, r.what); };
  int64_t lrows = 0;
  const std::vector<unsigned char> blob = plan_rhythm_files(p, &lrows);
  if (!blob.empty()) {
    if (!r.table(&b->d_rt_long, blob.data(), blob.size(), "rhythm long files")) return failed();
    // (magnitude, phase) pairs, then the follower's float per bin: 12 bytes per bin and frame
    if (!r(w.rt_polar, &b->d_rt_polar, (size_t)lrows * 256 * (sizeof(float2) + sizeof(float)), "rhythm polar")) return failed();
    b->rt_long_rows = lrows;
  }
  set_rhythm_context(b, nullptr);
  if (!r.table(&b->d_rt_files, b->rt_files.data(), b->rt_files.size() * sizeof(RhythmFile), "rhythm files")) return failed();
  b->rt_files_dirty = false;
  if (!r(w.rt_scalars, &b->d_rt_scalars, (size_t)n_bufs * AFX_NUM_RHYTHM_SCALARS * sizeof(double), "rhythm scalars")) return failed();
  const int64_t rows = b->rt_offset.back();
  if (rows > 0) {
    if (!r(w.rt_odf, &b->d_rt_odf, (size_t)rows * 2 * sizeof(float), "onset functions")) return failed();
    if (!r(w.rt_onsets, &b->d_rt_onsets, (size_t)rows * 2 * sizeof(double), "onsets")) return failed();
    if (!r(w.rt_scratch, &b->d_rt_scratch, (size_t)rows * 8 * sizeof(double), "rhythm scratch")) return failed();
  }
  if (p.want_stats) {
    if (!r.table(&b->d_rt_foff, b->rt_offset.data(), b->rt_offset.size() * sizeof(int64_t), "rhythm offsets")) return failed();
    if (!r(w.rt_stats, &b->d_rt_stats, (size_t)n_bufs * 2 * 13 * sizeof(double), "rhythm stats")) return failed();
  }
  return AFX_OK;
}

}  // namespace

// Common tail of batch creation (afx_batch_create: the caller's buffers; afx_batch_create_from_raw: the LoadSample
// front end's output).
int build_batch(afx_plan* plan, const BatchSource& src, afx_batch** out_batch) {
  afx_batch* b = new (std::nothrow) afx_batch();
  if (!b) {
    ws_release(plan, src.acquired);
    return fail(AFX_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
  Planner p{plan, b, src, src.mask & ~(uint32_t)AFX_D_STATISTICS, (src.mask & AFX_D_STATISTICS) != 0, 0, 0};
  p.fmask = frames_mask(p.mask);   // the kernels see the descriptor bits only
  p.esz = (src.dtype == kPcmF64) ? 8 : 4;
  b->plan = plan;
  plan->refs.fetch_add(1);
  b->mask = p.mask;
  b->n_bufs = src.n_bufs;
  b->lay = make_layout(p.mask);
  // flux, the sub-band descriptors and the whitened-spectrum neighbours are computed from the stored
  // magnitudes by later kernels
  b->mag_wanted = (p.mask & kNeedsMagnitudes) != 0;
  b->frame_offset.assign((size_t)src.n_bufs + 1, 0);
  b->buf_status = *src.status;
  b->pcm_dtype = src.dtype;
  if (src.scales) b->buf_scale = *src.scales;

  place_buffers(p);
  if (p.frames > 0x7FFFFF00LL) {
    delete b;
    plan->refs.fetch_sub(1);
    ws_release(plan, src.acquired);
    return fail(AFX_ERR_INVALID_ARG, "more than 2^31 frames in one batch");
  }
  choose_frame_kernel(p);
  cut_chunks(p, choose_chunk_frames(p));

  auto cleanup = [&](int st) { afx_batch_destroy(b); return st; };
  hipError_t e = hipSuccess;
  b->ws = src.acquired ? src.acquired : ws_acquire(plan, &e);
  if (!b->ws) return cleanup(hip_fail(e, "workspace"));
  Workspace& w = *b->ws;
  b->stream = w.stream; b->ev0 = w.ev0; b->ev1 = w.ev1;
  Reserver reserve{plan, b->ws, b->stream};
  if (p.arena > 0) {
    if (!reserve(w.pcm, &b->d_pcm, (size_t)p.arena * p.esz + 64, "pcm")) return cleanup(hip_fail(reserve.e, reserve.what));
    const int st = src.fill(b, src.fill_ctx);
    if (st != AFX_OK) return cleanup(st);
  }
  int st = reserve_frame_buffers(p, reserve);
  if (st == AFX_OK && (p.mask & AFX_D_RHYTHM) && src.n_bufs > 0) st = reserve_rhythm_buffers(p, reserve);
  if (st == AFX_OK && !reserve.flush()) st = hip_fail(reserve.e, reserve.what);
  if (st != AFX_OK) return cleanup(st);
  // the caller's PCM buffers (afx_batch_create) may go away when this returns; the tables uploaded above are the batch's own
  if (src.wait_for_uploads && (e = hipStreamSynchronize(b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipStreamSynchronize"));
  *out_batch = b;
  return AFX_OK;
}

}  // namespace host
}  // namespace afx
