// afec_amd/csrc/afx_host.h -- host side of libafx_hip.so, shared between its translation units (round 5: afx_capi.cpp,
// 2 100 lines with three functions of 140-340 lines, became these):
//
//   afx_plan.cpp        status texts, build info, the constant tables of a plan (the analogue of the TSampleAnalyser
//                       constructor, SampleAnalyser.cpp:162-198) and the plan's C-ABI entry points
//   afx_workspace.cpp   pooled device workspaces, the device work queues' bookkeeping (QueueBook), the host's waits, the
//                       plan's upload / download streams, page-locked host memory
//   afx_batch_plan.cpp  what a batch launches and on what: record layout, arena placement, kernel layout, chunk tables,
//                       the rhythm tracker's file table, device buffers (build_batch)
//   afx_batch_create.cpp  afx_batch_create, afx_batch_create_from_raw (the LoadSample front end, SampleAnalyser.cpp:484-718),
//                       afx_extract_batch
//   afx_batch_run.cpp   afx_batch_run: the kernels of one pass in stream order (SampleAnalyser.cpp:814-1048)
//   afx_batch_fetch.cpp results back to the host, batch information, afx_batch_destroy
//
// Nothing here computes a descriptor: every kernel lives in the .hip files (afx_internal.h declares their launchers).
#pragma once

#include "../../include/afx.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "afx_internal.h"

namespace afx {
namespace host {

// status + thread-local detail text (afx_last_error)
int fail(int status, const std::string& msg);
int hip_fail(hipError_t e, const char* what);
const char* last_error_text();
#define HIP_TRY(expr)                                              \
  do {                                                             \
    hipError_t e_ = (expr);                                        \
    if (e_ != hipSuccess) return ::afx::host::hip_fail(e_, #expr); \
  } while (0)

struct DeviceTables {
  void* win = nullptr;
  void* t1 = nullptr;
  void* t2 = nullptr;
  void* post = nullptr;
  void* melw = nullptr;
  double* dct = nullptr;
  // double-precision FFT tables for the time-domain neighbours (alias t1/t2/post in the f64 mode)
  void* t1_f64 = nullptr;
  void* t2_f64 = nullptr;
  void* post_f64 = nullptr;
  // half-wave kernels (afx_frames32.hip), always double
  void* win32 = nullptr;
  void* tw32 = nullptr;
  void* post32 = nullptr;
  void* melw32 = nullptr;
  // rhythm tracker (afx_rhythm.hip)
  double* rt_window = nullptr;
  double* rt_tw = nullptr;
  double* rt_ut = nullptr;
  double* rt_canny = nullptr;
  double* rt_rayleigh = nullptr;
  // sample-rate conversion (afx_resample.hip): uploaded by the first batch that holds a file at another rate
  float* rs_filter = nullptr;
  // 64 bytes that afx_plan_probe_device writes to (a probe must not allocate: after an out-of-memory failure an
  // allocation may fail on a device that is perfectly alive)
  void* probe = nullptr;
};

// The device work queues of one workspace (afx_internal.h: WorkQueue): one counter per kernel in device memory, NEVER reset
// between launches -- a launch advances its counter by exactly its number of items, and the host passes the counter's
// value at launch.  This type owns the host's record of those values and the one protocol around it: take() counts a
// launch before it is enqueued; when a run fails between the two the record and the device disagree, mark_failed() says
// so, and repair() drains the streams and starts both from zero before the next run.  Nothing else touches the counts.
class QueueBook {
 public:
  // the counters' device memory (kQueueSlots unsigneds, zeroed on `stream`); the record starts from zero with it
  void attach(unsigned* device_counters) {
    d_ = device_counters;
    for (unsigned& c : count_) c = 0;
    dirty_ = false;
  }
  bool attached() const { return d_ != nullptr; }
  unsigned* device() const { return d_; }
  // the queue of a launch of `items` items on counter `slot`: its base is the counter's value when the launch starts
  WorkQueue take(int slot, int items) {
    WorkQueue q{d_ ? d_ + slot : nullptr, count_[slot]};
    if (d_) count_[slot] += (unsigned)items;
    return q;
  }
  void mark_failed() { dirty_ = true; }
  bool needs_repair() const { return dirty_ && d_; }
  // everything enqueued so far is drained, then the counters and the record restart from zero
  hipError_t repair(hipStream_t stream, hipStream_t side_stream) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess && side_stream) e = hipStreamSynchronize(side_stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_, 0, kQueueSlots * sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
    for (unsigned& c : count_) c = 0;
    dirty_ = false;
    return hipSuccess;
  }

 private:
  unsigned* d_ = nullptr;
  unsigned count_[kQueueSlots] = {};
  bool dirty_ = false;
};

// Device buffers, stream and events of one batch.  Kept in a small per-plan pool so that the
// one-file-per-call pattern of the reference (one Extract() per worker thread and file,
// Crawler.cpp:706-728) does not pay hipMalloc / hipStreamCreate on every call.
struct Workspace {
  struct Buf {
    void* p = nullptr;
    size_t cap = 0;
  };
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // second stream for the rhythm tracker's kernels, which depend on the PCM only: small batches (a crawler's 256 files)
  // leave most of the chip idle during any one kernel, so the two kernel chains run side by side
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_time_join = nullptr;   // the time-domain kernels (autocorrelation, f0, hop) run on the side stream too
  hipEvent_t ev_copy = nullptr;   // the host's waits for the plan's upload / download streams (and, when blocking, for this batch's stream)
  bool blocking = false;          // the host's waits of this workspace's batches sleep (afx_plan_set_blocking_wait)
  QueueBook queues;               // the kernels' work-queue counters (their memory is `queue` below)
  Buf tables;                     // the host-filled tables of a batch, one block (afx_batch_plan.cpp: Reserver)
  void* h_pin = nullptr;          // page-locked host block the tables are uploaded from (and the scan results land in)
  size_t h_pin_cap = 0;
  Buf pcm, rec, mag, stats, follower, efflen, raw, files, scan, partial, place, queue;
  Buf rt_odf, rt_onsets, rt_scratch, rt_scalars, rt_stats, rt_polar;   // rhythm tracker (its host-filled tables lie in `tables`)
  Buf stat_tmp;                                                                 // half-wave statistics class
  Buf rs_files, rs_groups, rs_ngroups;                                          // sample-rate conversion (afx_resample.hip)
  std::vector<Buf*> all_bufs();   // every Buf member above, each exactly once: ws_free and bytes() walk this list
  size_t bytes();
};

}  // namespace host
}  // namespace afx

struct afx_plan {
  // the plan handle and every live batch hold one reference; the last one to go frees the plan (a batch destroyed
  // after afx_plan_destroy still finds its plan, its device and its workspace pool)
  std::atomic<int> refs{1};
  std::mutex pool_mutex;
  std::atomic<bool> blocking_wait{false};   // afx_plan_set_blocking_wait: the host's waits of this plan's batches sleep
  std::vector<afx::host::Workspace*> pool;
  afx_plan_desc desc;
  int first_bin, last_bin, bin_count;
  std::vector<double> window;  // [fft]
  std::vector<double> mel;     // [14][fft/2]
  afx::host::DeviceTables dev;
  int cu_count = 256;
  // The large transfers of every batch of this plan go through ONE upload stream and ONE download stream.  Measured
  // on the pool (tools/link_rate.py): one stream per direction runs full duplex at 46 + 46 GB/s, three batch streams
  // that each upload and download collapse to 16 + 16 GB/s (the copy engines are re-assigned back and forth).
  std::mutex up_mutex, down_mutex;
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  hipStream_t probe_stream = nullptr;   // afx_plan_probe_device: carries nothing else, so waiting for it waits for nobody's work
  // afx_plan_desc.frame_kernel: 0 = 64-lane frame kernels only (A/B timing), 1 = by batch size (default),
  // 2 = half-wave kernel for every batch it supports (tests)
  int halfwave = 1;
  // afx_plan_desc.flags & AFX_PLAN_NO_SIDE_STREAM: the rhythm tracker's kernels are enqueued on the batch's own stream
  // instead of its side stream (per-kernel durations of a profile are then not inflated by overlap)
  bool side_stream = true;
};

struct afx_batch {
  afx_plan* plan = nullptr;
  uint32_t mask = 0;
  int pcm_dtype = afx::kPcmF32;   // afx::kPcm*: the ABI's AFX_PCM_F32 / AFX_PCM_F64, or kPcmScaledF32 behind the LoadSample front end
  int32_t n_bufs = 0;
  std::vector<int64_t> frame_offset;  // [n_bufs+1]
  std::vector<int32_t> buf_status;    // [n_bufs]
  int64_t total_frames = 0;
  int n_chunks = 0;
  int grid_blocks = 0;
  afx::RecordLayout lay{};
  afx::host::Workspace* ws = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  void* d_pcm = nullptr;
  afx::Chunk* d_chunks = nullptr;
  afx::ChunkRemaining* d_rem = nullptr;
  int32_t* d_chunk_first = nullptr;      // whitening kernels: [n_bufs + 1] into d_wchunks
  afx::Chunk* d_wchunks = nullptr;       // whitening kernels' chunk table
  int n_wchunks = 0;
  bool need_follow = false;
  afx::BufSpan* d_spans = nullptr;
  int32_t* d_efflen = nullptr;
  double* d_follower = nullptr;
  int chunk_frames = 0;
  double* d_rec = nullptr;
  double* d_mag = nullptr;
  int64_t* d_frame_offset = nullptr;
  double* d_stats = nullptr;
  double* d_stat_tmp = nullptr;  // half-wave statistics class: raw sums per frame
  std::vector<int64_t> arena_off, used;  // per buffer: start and length (samples) of its analysed prefix in d_pcm
  std::vector<double> buf_scale;         // kPcmScaledF32: FinalScaling per buffer (empty otherwise)
  // rhythm tracker (AFX_D_RHYTHM)
  std::vector<int64_t> rt_offset;          // [n_bufs+1]: rows of the 512/128 frames
  std::vector<afx::RhythmFile> rt_files;   // [n_bufs]
  std::vector<int64_t> file_samples;       // [n_bufs]: mOriginalNumberOfSamples default (the buffer's / file's own length)
  std::vector<int32_t> file_offset;        // [n_bufs]: mDataOffset default
  std::vector<int32_t> file_rate;          // [n_bufs]: mOriginalSampleRate default (0: the plan's rate)
  bool rt_files_dirty = false;
  afx::RhythmFile* d_rt_files = nullptr;
  // long files of a small batch (afx_rhythm.hip): [n_long] file indices, [n_long + 1] round offsets (int32), [n_long + 1]
  // frame offsets (int64) in one device buffer; the polar scratch
  int32_t rt_n_long = 0, rt_long_rounds = 0;
  int64_t rt_long_rows = 0;
  void* d_rt_long = nullptr;
  float2* d_rt_polar = nullptr;
  float* d_rt_odf = nullptr;
  double* d_rt_onsets = nullptr;
  double* d_rt_scratch = nullptr;
  double* d_rt_scalars = nullptr;
  double* d_rt_stats = nullptr;
  int64_t* d_rt_foff = nullptr;
  // host copies of the tables that are uploaded asynchronously: they live as long as the batch, so creation does not
  // have to wait for the uploads (afx_batch_destroy synchronises the stream before they go)
  std::vector<afx::Chunk> h_chunks;
  std::vector<afx::ChunkRemaining> h_remaining;
  std::vector<int32_t> h_chunk_first;
  std::vector<afx::Chunk> h_wchunks;
  std::vector<afx::BufSpan> h_spans;
  std::vector<afx::LoadPlace> h_place;
  bool mag_wanted = false;
  bool ran = false;        // afx_batch_run has been enqueued at least once: the fetches have something to fetch
  bool halfwave = false;   // frames by the half-wave kernel: a wave walks two chunks at a time
};

namespace afx {
namespace host {

// ---- afx_plan.cpp ----
void plan_release(afx_plan* plan);                 // drops one reference; the last one frees the plan
hipError_t resample_filter_table(afx_plan* plan);  // the converter's filter, uploaded once per plan by the first batch that needs it
int64_t analysed_length(const afx_plan* p, int64_t n_samples);   // SampleDataAnalyzationLength, SampleAnalyser.cpp:760-764
int64_t num_frames(const afx_plan* p, int64_t n_samples);        // SampleAnalyser.cpp:814

// ---- afx_workspace.cpp ----
Workspace* ws_acquire(afx_plan* plan, hipError_t* err);
void ws_release(afx_plan* plan, Workspace* w);
void ws_free(Workspace* w);
// at least `bytes` in b.  When the device is out of memory the plan's idle pooled workspaces are freed and the
// allocation is tried once more (they hold their capacity: a batch that failed for memory would otherwise fail again)
hipError_t ws_reserve(afx_plan* plan, Workspace::Buf& b, size_t bytes);
hipError_t ws_pin_reserve(Workspace* w, size_t bytes);   // at least `bytes` of page-locked host memory in w->h_pin
size_t pool_trim(afx_plan* plan);   // frees every idle pooled workspace of the plan; returns the bytes given back
hipError_t wait_for_event(Workspace* ws, hipEvent_t ev);
hipError_t wait_for_stream(Workspace* ws, hipStream_t stream);
hipError_t upload_through_plan(afx_plan* plan, Workspace* ws, void* dst, const void* src, size_t bytes);
struct Download { void* dst; const void* src; size_t bytes; };
hipError_t download_through_plan(afx_batch* b, const Download* items, int n);

// ---- afx_batch_plan.cpp ----
// Every per-frame series of the ABI: where it lives in afx_out / afx_stats_out / the device record, its
// width, and the mask bit that selects it.  The order is the record order.
struct FieldDesc {
  double* afx_out::*out;
  double* afx_stats_out::*stat;
  int32_t RecordLayout::*off;
  int width;
  uint32_t bit;
};
extern const FieldDesc kFields[AFX_NUM_SERIES];
RecordLayout make_layout(uint32_t mask);
// which kernels a descriptor mask needs
constexpr uint32_t kSpectralBits = AFX_D_ALL_LOW_LEVEL | AFX_D_MAGNITUDE;
constexpr uint32_t kNeedsMagnitudes = AFX_D_MAGNITUDE | AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX |
                                      AFX_D_SPECTRAL_COMPLEXITY | AFX_D_F0;
constexpr uint32_t kWhitenBits = AFX_D_SPECTRAL_COMPLEXITY | AFX_D_F0 | AFX_D_SPECTRAL_INHARMONICITY | AFX_D_TRISTIMULUS;
constexpr uint32_t kTimeBits = AFX_D_AMPLITUDE_SILENCE | AFX_D_AMPLITUDE_ENVELOPE | AFX_D_AUTO_CORRELATION | AFX_D_F0;
uint32_t frames_mask(uint32_t mask);   // the bits the frame kernel sees
bool mask_ok(uint32_t mask);
void set_rhythm_context(afx_batch* b, const afx_file_info* info);

// What the two ways of creating a batch hand to build_batch: lengths[i] = samples of buffer i as
// AnalyzeLowLevelDescriptors would see them; fill() puts the analysed prefix of every buffer at arena_off[i] of
// b->d_pcm using b->stream (a plain function + context: the translation units do not share templates).
struct BatchSource {
  int32_t n_bufs = 0;
  uint32_t mask = 0;
  int dtype = kPcmF32;                              // afx::kPcm*
  const std::vector<int64_t>* lengths = nullptr;    // [n_bufs]
  const std::vector<int32_t>* status = nullptr;     // [n_bufs]
  int (*fill)(afx_batch* b, void* ctx) = nullptr;
  void* fill_ctx = nullptr;
  Workspace* acquired = nullptr;                    // a workspace the caller already holds (it moves into the batch; released on failure)
  const std::vector<int64_t>* file_samples = nullptr;   // rhythm tracker context of LoadSample batches
  const std::vector<int32_t>* file_offset = nullptr;
  const std::vector<int32_t>* file_rate = nullptr;
  const std::vector<double>* scales = nullptr;      // kPcmScaledF32: FinalScaling per buffer
  bool wait_for_uploads = true;
};
int build_batch(afx_plan* plan, const BatchSource& src, afx_batch** out_batch);

// ---- afx_batch_run.cpp ----
#if defined(AFX_STAMPS) && AFX_STAMPS
void stamps_report();   // diagnostic build only (make stamps)
#endif

// ---- afx_batch_create.cpp ----
int first_valid_dtype(const afx_buf* bufs, int32_t n_bufs);
int batch_create_typed(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask, int dtype, afx_batch** out_batch);

}  // namespace host
}  // namespace afx
