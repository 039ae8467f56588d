// afec_amd/csrc/afx_batch_run.cpp -- afx_batch_run: the kernels of one pass over a resident batch, in stream order --
// the body of the per-frame loop of TSampleAnalyser::AnalyzeLowLevelDescriptors (SampleAnalyser.cpp:814-976), the
// 512/128 rhythm tracker's loop behind it (983-1048) and CalcStatistics (1065).  See afx_host.h for the map of the host side.

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "afx_host.h"

namespace afx {
namespace host {

#if defined(AFX_STAMPS) && AFX_STAMPS
unsigned long long* g_stamp_buf = nullptr;
#endif

namespace {

// which of the two statistics kernels a set of series needs (afx_stats.hip)
void stats_regimes(const std::vector<int64_t>& offset, afx::StatsArgs* sa) {
  sa->small_rows = 0;
  sa->need_long = 0;
  for (size_t i = 0; i + 1 < offset.size(); ++i) {
    const int64_t n = offset[i + 1] - offset[i];
    if (n >= 2 && n <= 128) sa->small_rows = std::max<int32_t>(sa->small_rows, (int32_t)n);
    else sa->need_long = 1;
  }
}
void stats_regimes(const afx_batch* b, afx::StatsArgs* sa) { stats_regimes(b->frame_offset, sa); }

// the rhythm tracker's kernels + the statistics of the two onset series (TSampleAnalyser::CalcStatistics covers them)
int run_rhythm(afx_batch* b, hipStream_t stream) {
  if (!(b->mask & AFX_D_RHYTHM) || b->n_bufs == 0) return AFX_OK;
  const afx_plan* plan = b->plan;
  const float rate = (float)plan->desc.sample_rate;
  afx::RhythmArgs ra{};
  ra.pcm = b->d_pcm; ra.pcm_dtype = b->pcm_dtype; ra.n_files = b->n_bufs; ra.files = b->d_rt_files;
  ra.total_frames = b->rt_offset.back(); ra.sample_rate = plan->desc.sample_rate; ra.rayleigh_n = afx::kRayleighTable;
  ra.window = plan->dev.rt_window; ra.tw256 = plan->dev.rt_tw; ra.ut512 = plan->dev.rt_ut; ra.canny = plan->dev.rt_canny;
  ra.rayleigh = plan->dev.rt_rayleigh;
  // TOnsetFftProcessor::SetRelaxTime(25.0f) (OnsetDetector.cpp:105-112; MOnsetWhiteningRelaxTime, RhythmTracker.cpp:21)
  ra.relax_coef = (float)(std::exp((-2.30258509 * (float)128) / ((float)25.0 * rate)));
  ra.norm_complex = (float)(231.70475 / std::pow((double)512, 1.5));   // kFunctionRComplex, OnsetDetector.cpp:300-302
  ra.norm_power = 2560.f / (float)(257 * 512);                          // kFunctionPower, OnsetDetector.cpp:277-279
  ra.thresh[0] = (float)0.2; ra.thresh[1] = (float)0.8;                 // RhythmTracker.cpp:26, 32
  ra.medspan = std::max(3, (int)((rate * (float)0.2) / (float)128 + 0.5f));      // OnsetDetector.cpp:262-264
  ra.mingap[0] = (int)((rate * (float)0.06) / (float)128 + 0.5f);                 // OnsetDetector.cpp:272; RhythmTracker.cpp:28, 34
  ra.mingap[1] = (int)((rate * (float)0.12) / (float)128 + 0.5f);
  if (ra.medspan > 256) return fail(AFX_ERR_UNSUPPORTED, "sample rate too high for the onset detector's median span");
  for (const afx::RhythmFile& rf : b->rt_files)
    if (rf.frames <= afx::kRhythmLdsFrames) ra.lds_frames = std::max(ra.lds_frames, rf.frames);
  ra.odf = b->d_rt_odf; ra.onsets = b->d_rt_onsets; ra.scratch = b->d_rt_scratch; ra.scalars = b->d_rt_scalars;
  ra.n_long = b->rt_n_long; ra.long_rounds = b->rt_long_rounds;
  if (b->rt_n_long > 0) {
    const size_t nl = (size_t)b->rt_n_long;
    ra.long_frame_off = (const int64_t*)b->d_rt_long;
    ra.long_round_off = (const int32_t*)((const unsigned char*)b->d_rt_long + (nl + 1) * 8);
    ra.long_files = (const int32_t*)((const unsigned char*)b->d_rt_long + (nl + 1) * 12);
    ra.long_polar = b->d_rt_polar;
    ra.long_den = (float*)(b->d_rt_polar + b->rt_long_rows * 256);
  }
  HIP_TRY(afx::launch_rhythm(ra, stream));
  if (b->d_rt_stats) {
    afx::StatsArgs sa{};
    sa.rec = b->d_rt_onsets; sa.frame_offset = b->d_rt_foff; sa.n_bufs = b->n_bufs; sa.stride = 2;
    sa.stats = b->d_rt_stats;
    stats_regimes(b->rt_offset, &sa);
    HIP_TRY(afx::launch_stats(sa, stream));
  }
  return AFX_OK;
}

// the rhythm chain on the workspace's side stream, forked here and joined by rhythm_join at the end of the run
// (already_forked: the side stream waits for the batch's stream already -- the time-domain kernels went there first)
int rhythm_fork(afx_batch* b, bool already_forked) {
  if (!(b->mask & AFX_D_RHYTHM) || b->n_bufs == 0) return AFX_OK;
  Workspace& w = *b->ws;
  if (!b->plan->side_stream) return run_rhythm(b, b->stream);
  if (!already_forked) {
    HIP_TRY(hipEventRecord(w.ev_fork, b->stream));
    HIP_TRY(hipStreamWaitEvent(w.side_stream, w.ev_fork, 0));
  }
  const int st = run_rhythm(b, w.side_stream);
  if (st != AFX_OK) return st;
  HIP_TRY(hipEventRecord(w.ev_join, w.side_stream));
  return AFX_OK;
}
int rhythm_join(afx_batch* b) {
  if (!(b->mask & AFX_D_RHYTHM) || b->n_bufs == 0 || !b->plan->side_stream) return AFX_OK;
  HIP_TRY(hipStreamWaitEvent(b->stream, b->ws->ev_join, 0));
  return AFX_OK;
}

int launch_statistics(afx_batch* b) {
  if (!b->d_stats) return AFX_OK;
  StatsArgs sa{};
  sa.rec = b->d_rec; sa.frame_offset = b->d_frame_offset; sa.n_bufs = b->n_bufs; sa.stride = b->lay.stride;
  sa.stats = b->d_stats;
  stats_regimes(b, &sa);
  HIP_TRY(launch_stats(sa, b->stream));
  return AFX_OK;
}

// The time-domain kernels (autocorrelation, f0, the hop's descriptors) read the PCM only and the spectral chain
// (STFT, bands) does not need them: they go to the side stream, ahead of the rhythm tracker's, and are joined where
// the whitening kernel reads what the pitch kernel left (f0, its confidence, the hop's silence flag).  One kernel at a
// time leaves the tail of every launch to a partly idle chip (a crawl's batch is 10-40 waves per SIMD per kernel).
// *on_side: they were enqueued on the side stream (the caller joins them).
int launch_time_domain(afx_batch* b, bool* on_side) {
  *on_side = false;
  const DeviceTables& t = b->plan->dev;
  // the half-wave full classes (2, 3) leave the amplitude of the hop to hop_kernel / pitch_kernel; the magnitude class (4)
  // has the hop in registers at the top of a frame and writes it itself
  const int frame_class = b->halfwave ? frames32_class(frames_mask(b->mask)) : -1;
  const uint32_t post_amplitude = (b->total_frames > 0 && (frame_class == 2 || frame_class == 3))
                                      ? (b->mask & (AFX_D_AMPLITUDE_PEAK | AFX_D_AMPLITUDE_RMS)) : 0u;
  if (!(b->total_frames > 0 && ((b->mask & kTimeBits) || post_amplitude))) return AFX_OK;
  // (not with the rhythm tracker selected: its chain, the longer one, has the side stream then -- 21.7 against 21.2 M frames/s
  // on the C4 share with everything selected; without it 34.4 against 33.9 M on C3)
  const bool time_side = b->plan->side_stream && frames_mask(b->mask) != 0 && !(b->mask & AFX_D_RHYTHM);
  hipStream_t ts = b->stream;
  if (time_side) {
    HIP_TRY(hipEventRecord(b->ws->ev_fork, b->stream));
    HIP_TRY(hipStreamWaitEvent(b->ws->side_stream, b->ws->ev_fork, 0));
    ts = b->ws->side_stream;
  }
  TimeArgs ta{};
  ta.pcm = b->d_pcm; ta.chunks = b->d_chunks; ta.remaining = b->d_rem; ta.n_chunks = b->n_chunks;
  ta.pcm_dtype = b->pcm_dtype; ta.rec = b->d_rec; ta.lay = b->lay;
  ta.t1 = t.t1_f64; ta.t2 = t.t2_f64; ta.post = t.post_f64;
  ta.amplitude = post_amplitude;
  // with f0 selected the pitch kernel has the hop's samples in registers anyway and writes its descriptors too
  const bool want_hop = (b->mask & (AFX_D_AMPLITUDE_SILENCE | AFX_D_AMPLITUDE_ENVELOPE)) || post_amplitude;
  ta.hop_here = (want_hop && (b->mask & AFX_D_F0)) ? 1u : 0u;
  if (want_hop && !ta.hop_here) HIP_TRY(launch_hop(ta, ts));
  if (b->mask & AFX_D_F0) {
    ta.queue = b->ws->queues.take(kQueuePitch, b->n_chunks);
    HIP_TRY(launch_pitch(ta, ts));
  }
  if (b->mask & AFX_D_AUTO_CORRELATION) {
    ta.queue = b->ws->queues.take(kQueueAcorr, b->n_chunks);
    HIP_TRY(launch_acorr(ta, ts));
  }
  if (time_side) HIP_TRY(hipEventRecord(b->ws->ev_time_join, ts));
  *on_side = time_side;
  return AFX_OK;
}

// window -> FFT -> magnitude -> MFCC (+ what the batch's class of the frame kernel computes itself), SampleAnalyser.cpp:826-862
int launch_frame_kernel(afx_batch* b) {
  if (!frames_mask(b->mask)) return AFX_OK;
  const DeviceTables& t = b->plan->dev;
  FrameArgs a{};
  a.pcm = b->d_pcm;
  a.chunks = b->d_chunks;
  a.n_chunks = b->n_chunks;
  a.mask = frames_mask(b->mask);
  a.rec = b->d_rec;
  a.lay = b->lay;
  a.mag_out = b->d_mag;
  a.win = t.win; a.t1 = t.t1; a.t2 = t.t2; a.post = t.post; a.melw = t.melw; a.dct = t.dct;
  a.win32 = t.win32; a.tw32 = t.tw32; a.post32 = t.post32; a.melw32 = t.melw32;
#if defined(AFX_STAMPS) && AFX_STAMPS
  // diagnostic build: per-stage cycle counters of the half-wave kernel, printed when the batch is destroyed
  if (!g_stamp_buf) { HIP_TRY(hipMalloc((void**)&g_stamp_buf, (16 + 8192) * 8)); HIP_TRY(hipMemset(g_stamp_buf, 0, (16 + 8192) * 8)); }
  a.stamps = g_stamp_buf;
#endif
  if (!b->halfwave) {
    HIP_TRY(launch_frames(a, b->plan->desc.precision, b->pcm_dtype, b->grid_blocks, b->stream));
    return AFX_OK;
  }
  const WorkQueue q = b->ws->queues.take(kQueueFrames32, (b->n_chunks + 1) / 2);   // a wave draws chunk PAIRS
  a.queue = q.counter;
  a.queue_base = q.base;
  a.stat_tmp = b->d_stat_tmp;
  a.mag_spare_row = b->total_frames;
  HIP_TRY(launch_frames32(a, b->grid_blocks, b->stream, b->total_frames, b->pcm_dtype == kPcmScaledF32));
  return AFX_OK;
}

// flux, the sub-band descriptors (SampleAnalyser.cpp:2067-2308) and, behind the half-wave frame kernel, the 28 spectrum
// bands and the spectral statistics' raw sums, from the stored magnitudes
int launch_band_kernel(afx_batch* b) {
  // the half-wave full class leaves the 28 spectrum bands to bands_kernel (it stores the magnitudes for it)
  const bool bands28_later = b->halfwave && (b->mask & AFX_D_SPECTRUM_BANDS);
  // ... and its magnitude class the spectral statistics
  const bool stats_later = b->halfwave && frames32_class(frames_mask(b->mask)) == 4;
  if (!((b->mask & (AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX)) || bands28_later)) return AFX_OK;
  BandArgs ba{};
  ba.mag = b->d_mag; ba.chunks = b->d_chunks; ba.n_chunks = b->n_chunks; ba.rec = b->d_rec; ba.lay = b->lay;
  ba.flags = ((b->mask & AFX_D_BAND_FEATURES) ? kBandsFeatures : 0) | ((b->mask & AFX_D_SPECTRAL_FLUX) ? kBandsFlux : 0) |
             (bands28_later ? kBandsSpectrum : 0) | (stats_later ? kBandsStats : 0);
  ba.stat_tmp = b->d_stat_tmp;
  ba.queue = b->ws->queues.take(kQueueBands, b->n_chunks);
  HIP_TRY(launch_bands(ba, b->stream));
  if (stats_later) {
    FrameArgs fa{};
    fa.mask = frames_mask(b->mask); fa.rec = b->d_rec; fa.lay = b->lay; fa.stat_tmp = b->d_stat_tmp;
    HIP_TRY(launch_stats32_finish(fa, b->stream, b->total_frames));
  }
  return AFX_OK;
}

// adaptive whitening -> peak spectrum -> spectral complexity, fail-safe f0 (SampleAnalyser.cpp:850-862, 876-927)
int launch_whitening(afx_batch* b) {
  if (!(b->mask & kWhitenBits)) return AFX_OK;
  WhitenArgs wa{};
  wa.mag = b->d_mag; wa.frame_offset = b->d_frame_offset; wa.n_bufs = b->n_bufs; wa.mask = b->mask;
  wa.chunk_first = b->d_chunk_first; wa.chunks = b->d_wchunks; wa.n_chunks = b->n_wchunks;
  wa.chunk_frames = b->chunk_frames; wa.follower = b->d_follower; wa.need_follow = b->need_follow ? 1 : 0;
  wa.rec = b->d_rec; wa.lay = b->lay;
  // new_aubio_spectral_whitening + set_relax_time(MSpectralWhiteningDecay = 22): awhitening.c:53-87,
  // SampleAnalyser.cpp:44, 805-809
  wa.decay = std::pow(0.001, (double)((float)b->plan->desc.hop_size / (float)b->plan->desc.sample_rate) / 22.0);
  wa.floor_value = 1.e-4;
  wa.queue = b->ws->queues.take(kQueueWhiten, b->n_wchunks);
  HIP_TRY(launch_whiten(wa, b->stream));
  return AFX_OK;
}

int batch_run_enqueue(afx_batch* b) {
  b->ran = true;
  bool time_side = false;
  int st = launch_time_domain(b, &time_side);
  if (st != AFX_OK) return st;
  if ((st = rhythm_fork(b, time_side)) != AFX_OK) return st;
  if (b->d_efflen) {
    // DbToLin(-48 / -24 / -12), AudioMath.inl:108-123
    const double k = std::log(10.0) / 20.0;
    HIP_TRY(launch_effective_length(b->d_pcm, b->pcm_dtype, b->d_spans, b->n_bufs, std::exp(-48.0 * k), std::exp(-24.0 * k),
                                    std::exp(-12.0 * k), b->d_efflen, b->stream));
  }
  if (b->total_frames == 0) {
    // nothing to analyse; empty series still reduce to TStatistics::Calc's Length == 0 result
    if ((st = launch_statistics(b)) != AFX_OK) return st;
    return rhythm_join(b);
  }
  if ((st = launch_frame_kernel(b)) != AFX_OK) return st;
  if ((st = launch_band_kernel(b)) != AFX_OK) return st;
  if (time_side) HIP_TRY(hipStreamWaitEvent(b->stream, b->ws->ev_time_join, 0));
  if ((st = launch_whitening(b)) != AFX_OK) return st;
  if ((st = launch_statistics(b)) != AFX_OK) return st;
  return rhythm_join(b);
}

}  // namespace

#if defined(AFX_STAMPS) && AFX_STAMPS
// diagnostic build only (make stamps): per-stage cycle counters of the half-wave kernel, printed when a batch is destroyed
void stamps_report() {
  if (!g_stamp_buf) return;
  unsigned long long h[16];
  if (hipMemcpy(h, g_stamp_buf, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[15]) {
    static const char* names[10] = {"wait DMA+window", "hop+convert", "P1", "exchange", "DMA issue+twiddle", "P2",
                                    "untangle", "mel rows", "win issue+reduce", "log+DCT"};
    unsigned long long tot = 0;
    for (int i = 0; i < 10; ++i) tot += h[i];
    std::fprintf(stderr, "[afx stamps] %llu stamped iterations (2 frames each), %.0f cycles per iteration\n", h[15], (double)tot / h[15]);
    for (int i = 0; i < 10; ++i) std::fprintf(stderr, "[afx stamps] %-18s %8.0f cycles  %5.1f %%\n", names[i], (double)h[i] / h[15], 100.0 * h[i] / tot);
    // lifetimes of every wave of the LAST launch, by wave index inside the workgroup
    static unsigned long long life[8192];
    if (hipMemcpy(life, g_stamp_buf + 16, sizeof life, hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long r0 = ~0ull, r1 = 0;
      double sum_core = 0, sum_real = 0;
      double wsum[8] = {}, wmin[8], wmax[8] = {}, wend[8] = {};
      int wn[8] = {};
      for (int w = 0; w < 8; ++w) wmin[w] = 1e30;
      for (int i = 0; i < 2048; ++i) {
        if (!life[4 * i + 1]) continue;
        r0 = std::min(r0, life[4 * i + 2]); r1 = std::max(r1, life[4 * i + 3]);
      }
      for (int i = 0; i < 2048; ++i) {
        if (!life[4 * i + 1]) continue;
        const int w = i & 7;
        const double dr = (double)(life[4 * i + 3] - life[4 * i + 2]) / 100.0;
        sum_core += (double)(life[4 * i + 1] - life[4 * i]); sum_real += dr * 100.0;
        wsum[w] += dr; wmin[w] = std::min(wmin[w], dr); wmax[w] = std::max(wmax[w], dr); ++wn[w];
        wend[w] += (double)(life[4 * i + 3] - r0) / 100.0;
      }
      std::fprintf(stderr, "[afx stamps] last launch: span first-entry..last-exit %.1f us, core clock %.3f GHz\n", (double)(r1 - r0) / 100.0, sum_core / sum_real * 0.1);
      for (int w = 0; w < 8; ++w)
        if (wn[w]) std::fprintf(stderr, "[afx stamps]   wave %d: life mean %.1f us (min %.1f, max %.1f), mean exit at %.1f us\n", w, wsum[w] / wn[w], wmin[w], wmax[w], wend[w] / wn[w]);
    }
    hipMemset(g_stamp_buf, 0, (16 + 8192) * 8);
  }
}
#endif

}  // namespace host
}  // namespace afx

using namespace afx::host;

extern "C" {

int afx_batch_run(afx_batch* b) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  Workspace* const ws = b->ws;
  // a run that failed part-way left the work-queue counters and the host's record of them apart (a launch that was
  // counted and never ran): QueueBook::repair drains everything of that run, then both start from zero again
  if (ws && ws->queues.needs_repair()) HIP_TRY(ws->queues.repair(b->stream, ws->side_stream));
  const int st = batch_run_enqueue(b);
  if (st != AFX_OK && ws) ws->queues.mark_failed();
  return st;
}

int afx_batch_sync(afx_batch* b) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  HIP_TRY(hipStreamSynchronize(b->stream));
  return AFX_OK;
}

int afx_batch_run_timed(afx_batch* b, int32_t steps, float* elapsed_ms) {
  if (!b || steps < 1 || !elapsed_ms) return fail(AFX_ERR_INVALID_ARG, "bad argument");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  for (int i = 0; i < steps; ++i) {
    const int st = afx_batch_run(b);
    if (st != AFX_OK) return st;
  }
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  HIP_TRY(hipEventSynchronize(b->ev1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
  return AFX_OK;
}

}  // extern "C"
