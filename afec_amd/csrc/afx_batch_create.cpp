// afec_amd/csrc/afx_batch_create.cpp -- the ways a batch comes to be: the caller's normalised buffers
// (afx_batch_create), decoded files through the LoadSample front end on the GPU (afx_batch_create_from_raw,
// SampleAnalyser.cpp:484-718), and the one-call form (afx_extract_batch).  See afx_host.h for the map of the host side.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "afx_host.h"

namespace afx {
namespace host {

namespace {

// AFX_TIMING=1 in the environment: wall time of the phases of afx_batch_create_from_raw, summed over calls and printed
// when the process ends (diagnostic for pipelines; three clock reads per call otherwise)
struct CreateTiming {
  std::atomic<long long> ns[4]{};   // upload + scan + wait, host placement, build + LoadSample write + wait, calls
  bool on = std::getenv("AFX_TIMING") != nullptr;
  ~CreateTiming() {
    if (on && ns[3].load())
      std::fprintf(stderr, "[afx timing] create_from_raw x%lld: upload + scan + wait %.1f ms, placement %.1f ms, build + write + wait %.1f ms\n",
                   ns[3].load(), ns[0].load() * 1e-6, ns[1].load() * 1e-6, ns[2].load() * 1e-6);
  }
};
CreateTiming g_create_timing;

inline long long now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// bytes per sample of the decoded PCM formats (0: unknown format)
int raw_bytes_per_sample(int format) {
  switch (format) {
    case AFX_RAW_I16: return 2;
    case AFX_RAW_I24: return 3;
    case AFX_RAW_F32: case AFX_RAW_I32: return 4;
    case AFX_RAW_F64: return 8;
    default: return 0;
  }
}

struct CallerBuffers {   // fill() of afx_batch_create: the caller's PCM, one transfer per buffer
  const afx_buf* bufs;
  int32_t n_bufs;
  size_t esz;
};
int fill_from_caller(afx_batch* b, void* ctx) {
  const CallerBuffers& c = *(const CallerBuffers*)ctx;
  for (int i = 0; i < c.n_bufs; ++i)
    if (b->used[i] > 0) {
      hipError_t e = hipMemcpyAsync((char*)b->d_pcm + (size_t)b->arena_off[i] * c.esz, c.bufs[i].pcm,
                                    (size_t)b->used[i] * c.esz, hipMemcpyHostToDevice, b->stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpy(pcm)");
    }
  return AFX_OK;
}

}  // namespace

// first valid buffer's PCM type (the arena of a call is homogeneous); -1 when there is none
int first_valid_dtype(const afx_buf* bufs, int32_t n_bufs) {
  for (int i = 0; i < n_bufs; ++i) {
    const afx_buf& s = bufs[i];
    if (s.n_samples >= 0 && (s.n_samples == 0 || s.pcm) && (s.dtype == AFX_PCM_F32 || s.dtype == AFX_PCM_F64)) return s.dtype;
  }
  return -1;
}

// afx_batch_create with the PCM type of the whole call given (afx_extract_batch cuts large calls into groups: the
// type is decided once, over all buffers, not per group)
int batch_create_typed(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask, int dtype, afx_batch** out_batch) {
  if (!plan || !out_batch || n_bufs < 0 || (n_bufs > 0 && !bufs))
    return fail(AFX_ERR_INVALID_ARG, "null argument");
  *out_batch = nullptr;
  if (!mask_ok(mask)) return fail(AFX_ERR_INVALID_ARG, "bad descriptor mask");
  HIP_TRY(hipSetDevice(plan->desc.device));

  // one PCM dtype per batch (the arena is homogeneous); first valid buffer of the call decides
  std::vector<int32_t> status((size_t)n_bufs, AFX_OK);
  std::vector<int64_t> lengths((size_t)n_bufs, 0);
  for (int i = 0; i < n_bufs; ++i) {
    const afx_buf& s = bufs[i];
    const bool good = s.n_samples >= 0 && (s.n_samples == 0 || s.pcm) &&
                      (s.dtype == AFX_PCM_F32 || s.dtype == AFX_PCM_F64);
    if (!good) { status[i] = AFX_ERR_BAD_BUFFER; continue; }
    if (dtype < 0) dtype = s.dtype;
    if (s.dtype != dtype) { status[i] = AFX_ERR_BAD_BUFFER; continue; }
    lengths[i] = s.n_samples;
  }
  if (dtype < 0) dtype = AFX_PCM_F32;
  CallerBuffers caller{bufs, n_bufs, (dtype == AFX_PCM_F64) ? (size_t)8 : (size_t)4};
  BatchSource src;
  src.n_bufs = n_bufs; src.mask = mask; src.dtype = dtype; src.lengths = &lengths; src.status = &status;
  src.fill = fill_from_caller; src.fill_ctx = &caller;
  return build_batch(plan, src, out_batch);
}

namespace {

// The steps of afx_batch_create_from_raw share this.
struct RawBatch {
  afx_plan* plan;
  const afx_raw* raws;
  int32_t n_bufs;
  std::vector<int32_t> status;          // [n_bufs]
  std::vector<LoadFile> files;          // [n_bufs]: where each file's decoded PCM lies in the raw arena
  std::vector<ResampleFile> conv;       // files at another rate than the plan's: converted on the GPU first
  std::vector<int32_t> conv_buf;        // their buffer indices
  std::vector<int32_t> file_rate;       // [n_bufs]: 0 = the plan's
  std::vector<LoadScan> scan;           // [n_bufs]: the scan kernels' results
  std::vector<LoadPlace> place;         // [n_bufs]
  std::vector<int64_t> lengths;         // [n_bufs]: samples of the normalised, trimmed, padded buffer
  int64_t raw_bytes = 0, conv_bytes = 0, group_slots = 0, conv_blocks = 0, conv_max_in = 0;
  Workspace* ws = nullptr;
  unsigned char* d_raw = nullptr;
  LoadFile* d_files = nullptr;
};

// SampleAnalyser.cpp:472-482: 1..8 channels, non-empty; lays the files out back to back, each on a 16-byte boundary
void lay_out_raw_arena(RawBatch& r) {
  for (int i = 0; i < r.n_bufs; ++i) {
    const afx_raw& f = r.raws[i];
    const int bps = raw_bytes_per_sample(f.format);
    const bool good = bps && f.data && f.n_frames > 0 && f.n_frames < 0x7FFFFFFF && f.channels >= 1 && f.channels <= 8 && f.sample_rate >= 0;
    r.files[(size_t)i] = LoadFile{0, 0, 0, 0};
    if (!good) { r.status[(size_t)i] = AFX_ERR_BAD_BUFFER; continue; }
    r.files[(size_t)i] = LoadFile{r.raw_bytes, f.n_frames, f.channels, f.format};
    r.raw_bytes += ((int64_t)f.n_frames * f.channels * bps + 15) & ~(int64_t)15;
  }
}

// Sample-rate conversion (SampleAnalyser.cpp:563-607): Speed = file rate / analyser rate in double, the converter runs
// at factor 1 / Speed and fills NewSizeInSamples = max(1, d2iRound(n / Speed)) samples.  The mono mix and the converted
// samples of such a file live behind the decoded PCM of the batch in the raw arena; the LoadSample kernels then read
// the converted samples as a mono file of "16-bit floats".
void plan_conversions(RawBatch& r) {
  // What is refused, for that buffer only (AFX_ERR_UNSUPPORTED -> a "Sample failed to load" row; the reference would
  // convert these too): a file above 16 x the analyser's rate (705.6 kHz: outside what the kernels' zero margins cover),
  // and a file whose conversion yields 2^30 samples or more (6.8 hours at 44.1 kHz, 4 GiB of floats; the kernels index a
  // file's samples with 32 bits) -- a header that claims 1 Hz makes a 96 KB file 2^31 samples.  Until round 5 the bounds
  // were a rate below the analyser's / 64 and 2^28 samples: a two-hour 48 kHz recording became a failed row.  The
  // crawler cuts batches by the converted size (TCrawlOptions::mDeviceBytesPerBatch), so a long file travels alone.
  constexpr double kMaxConvertedSamples = 1073741824.0;
  const afx_plan* plan = r.plan;
  for (int i = 0; i < r.n_bufs; ++i) {
    const afx_raw& f = r.raws[i];
    if (r.status[(size_t)i] != AFX_OK || f.sample_rate == 0 || f.sample_rate == plan->desc.sample_rate) continue;
    r.file_rate[(size_t)i] = f.sample_rate;
    const double speed = (double)f.sample_rate / (double)plan->desc.sample_rate;
    if (speed == 1.0) continue;
    const double factor = 1.0 / speed, scaled = (double)(int)f.n_frames / speed;
    const double n_out_d = std::floor(scaled + 0.5);               // TMath::d2iRound of a positive value (InlineMath.inl:823-826)
    if (n_out_d >= kMaxConvertedSamples || factor < 1.0 / 16.0 || r.conv_blocks > 0x7FFFFFF0) {
      r.status[(size_t)i] = AFX_ERR_UNSUPPORTED;
      r.files[(size_t)i] = LoadFile{0, 0, 0, 0};
      continue;
    }
    ResampleFile c{};
    c.raw_off = r.files[(size_t)i].raw_off; c.n_in = f.n_frames; c.channels = f.channels; c.format = f.format; c.factor = factor;
    c.n_out = std::max<int64_t>(1, (int64_t)n_out_d);
    c.mono_off = r.raw_bytes + r.conv_bytes;
    r.conv_bytes += ((c.n_in + 2 * kResampleMargin) * 4 + 15) & ~(int64_t)15;
    c.out_off = r.raw_bytes + r.conv_bytes;
    r.conv_bytes += (c.n_out * 4 + 15) & ~(int64_t)15;
    c.group_off = r.group_slots;                      // one record per 16 output samples
    r.group_slots += (c.n_out + 15) / 16;
    c.block_off = r.conv_blocks;
    r.conv_blocks += resample_blocks(c.n_out);
    r.conv_max_in = std::max(r.conv_max_in, c.n_in);
    r.conv.push_back(c);
    r.conv_buf.push_back(i);
  }
}

// the decoded PCM into the workspace's raw arena
hipError_t upload_raw(RawBatch& r, const char** what) {
  // Files that lie back to back in host memory, each starting at the next 16-byte boundary (a pipeline's staging
  // buffer), have the layout of the device arena: one transfer moves them all.  Otherwise one transfer per file
  // (each costs ~10 us of runtime overhead, which dominates for thousands of short files).
  *what = "hipMemcpy(raw)";
  bool contiguous = true;
  const char* base = nullptr;
  for (int i = 0; i < r.n_bufs && contiguous; ++i)
    if (r.status[(size_t)i] == AFX_OK) {
      if (!base) base = (const char*)r.raws[i].data - r.files[(size_t)i].raw_off;
      contiguous = ((const char*)r.raws[i].data == base + r.files[(size_t)i].raw_off);
    }
  if (contiguous && base) {
    // from the first valid file's first byte up to the last valid file's real end: raw_bytes rounds every file up to 16
    // bytes, and the bytes behind the caller's last buffer are not the library's to read (a buffer may end at the end of
    // a mapping) -- nor are the bytes in front of the first valid one: a file that plan_conversions refused keeps its
    // place in the arena's layout, and `base` then lies that far in front of the caller's memory (found by
    // tests/sanitize/fuzz_host_abi.cpp: a refused first file + one valid scattered buffer read 300 KB before it)
    int64_t first = -1, used = 0;
    for (int i = 0; i < r.n_bufs; ++i)
      if (r.status[(size_t)i] == AFX_OK) {
        if (first < 0) first = r.files[(size_t)i].raw_off;
        used = r.files[(size_t)i].raw_off + (int64_t)r.raws[i].n_frames * r.raws[i].channels * raw_bytes_per_sample(r.raws[i].format);
      }
    return upload_through_plan(r.plan, r.ws, r.d_raw + first, base + first, (size_t)(used - first));
  }
  for (int i = 0; i < r.n_bufs; ++i)
    if (r.status[(size_t)i] == AFX_OK) {
      const int bps = raw_bytes_per_sample(r.raws[i].format);
      const hipError_t e = hipMemcpyAsync(r.d_raw + r.files[(size_t)i].raw_off, r.raws[i].data,
                                          (size_t)r.raws[i].n_frames * r.raws[i].channels * bps, hipMemcpyHostToDevice, r.ws->stream);
      if (e != hipSuccess) return e;
    }
  return hipSuccess;
}

// upload -> (conversion) -> the scan kernels (peak, rms, first / last sample above the floor) -> results on the host
hipError_t stage_and_scan(RawBatch& r, const char** what) {
  afx_plan* plan = r.plan;
  Workspace* ws = r.ws;
  hipStream_t s = ws->stream;
  const int n_bufs = r.n_bufs;
  hipError_t e;
  *what = "hipMalloc(raw staging)";
  if ((e = ws_reserve(plan, ws->raw, (size_t)(r.raw_bytes + r.conv_bytes) + 16)) != hipSuccess) return e;
  if ((e = ws_reserve(plan, ws->files, r.files.size() * sizeof(LoadFile))) != hipSuccess) return e;
  if ((e = ws_reserve(plan, ws->scan, r.scan.size() * sizeof(LoadScan))) != hipSuccess) return e;
  if ((e = ws_reserve(plan, ws->partial, (size_t)n_bufs * load_scan_blocks_per_file(n_bufs) * 16)) != hipSuccess) return e;
  r.d_raw = (unsigned char*)ws->raw.p;
  r.d_files = (LoadFile*)ws->files.p;
  LoadScan* d_scan = (LoadScan*)ws->scan.p;
  if ((e = upload_raw(r, what)) != hipSuccess) return e;
  if (!r.conv.empty()) {
    *what = "resample";
    if ((e = resample_filter_table(plan)) != hipSuccess) return e;
    if ((e = ws_reserve(plan, ws->rs_files, r.conv.size() * sizeof(ResampleFile))) != hipSuccess) return e;
    if ((e = ws_reserve(plan, ws->rs_groups, (size_t)r.group_slots * sizeof(ResampleGroup))) != hipSuccess) return e;
    // (pageable source: the copy has left `conv` when the scan below has been waited for)
    if ((e = hipMemcpyAsync(ws->rs_files.p, r.conv.data(), r.conv.size() * sizeof(ResampleFile), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    if ((e = launch_resample(r.d_raw, (const ResampleFile*)ws->rs_files.p, (int)r.conv.size(), r.conv_blocks, r.conv_max_in,
                             (ResampleGroup*)ws->rs_groups.p, plan->dev.rs_filter, s)) != hipSuccess) return e;
    for (size_t k = 0; k < r.conv.size(); ++k)
      r.files[(size_t)r.conv_buf[k]] = LoadFile{r.conv[k].out_off, r.conv[k].n_out, 1, kRawMonoFloat};
  }
  *what = "load_scan";
  // the file table up and the scan results down through the workspace's page-locked block (pageable copies are staged
  // by the runtime on this thread and complete through its event thread: host CPU a crawl's workers do not have to spend)
  const size_t files_bytes = r.files.size() * sizeof(LoadFile), scan_bytes = r.scan.size() * sizeof(LoadScan);
  const size_t scan_off = (files_bytes + 63) & ~(size_t)63;
  if ((e = ws_pin_reserve(ws, scan_off + scan_bytes)) != hipSuccess) return e;
  std::memcpy(ws->h_pin, r.files.data(), files_bytes);
  if ((e = hipMemcpyAsync(r.d_files, ws->h_pin, files_bytes, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
  // -48 dB of full scale (MSilenceThresholdDb, SampleAnalyser.cpp:51, 648-649)
  const double silence_floor = 32768.0 * std::exp(-48.0 * (std::log(10.0) / 20.0));
  if ((e = launch_load_scan(r.d_raw, r.d_files, n_bufs, silence_floor, ws->partial.p, d_scan, s)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync((unsigned char*)ws->h_pin + scan_off, d_scan, scan_bytes, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  *what = "hipStreamSynchronize";
  if ((e = wait_for_stream(ws, s)) != hipSuccess) return e;
  std::memcpy(r.scan.data(), (const unsigned char*)ws->h_pin + scan_off, scan_bytes);
  return hipSuccess;
}

// padding rules of SampleAnalyser.cpp:681-701: where the audible part of every file goes and how long its buffer is
void place_files(RawBatch& r, afx_load_info* info) {
  const int fft = r.plan->desc.fft_size;
  for (int i = 0; i < r.n_bufs; ++i) {
    r.place[(size_t)i] = LoadPlace{};
    if (r.status[(size_t)i] != AFX_OK) { if (info) info[i] = afx_load_info{}; continue; }
    // silent leading samples = index of the first sample above the floor (all of them when there is none); the
    // trailing scan stops above that sample (SampleAnalyser.cpp:651-669)
    const LoadScan& sc = r.scan[(size_t)i];
    const int64_t n = r.files[(size_t)i].n_frames;
    const int64_t lead = (sc.trail < 0) ? n : sc.lead, trail = (sc.trail < 0) ? 0 : n - 1 - sc.trail;
    const int64_t audible = n - lead - trail;
    const int64_t end_pad = ((audible % fft) < fft / 2) ? fft / 2 : 0;
    const int64_t start_pad = (audible + end_pad < fft) ? fft - audible - end_pad : 0;
    r.lengths[(size_t)i] = audible + start_pad + end_pad;
    LoadPlace& pl = r.place[(size_t)i];
    pl.lead = lead; pl.audible = audible; pl.start_pad = start_pad;
    pl.scaling = sc.amplification / 32768.0;      // FinalScaling, SampleAnalyser.cpp:712
    if (info) {
      info[i].peak_value = (float)std::min(1.0, (double)sc.max_amp / 32768.0);
      info[i].rms_value = (float)std::min(1.0, std::sqrt(sc.sum_sq / (double)n));
      info[i].data_offset = (int32_t)(-lead + start_pad);
      info[i].silent_leading = (int32_t)lead;
      info[i].silent_trailing = (int32_t)trail;
      info[i].reserved = 0;
      info[i].n_samples = r.lengths[(size_t)i];
    }
  }
}

// fill() of the LoadSample batches: the write kernel stores the float mono signal behind the start pad
int fill_from_staging(afx_batch* b, void* ctx) {
  RawBatch& r = *(RawBatch*)ctx;
  for (int i = 0; i < r.n_bufs; ++i) r.place[(size_t)i].out_off = b->arena_off[i], r.place[(size_t)i].out_n = b->used[i];
  hipError_t e = ws_reserve(r.plan, r.ws->place, r.place.size() * sizeof(LoadPlace));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(place)");
  LoadPlace* d_place = (LoadPlace*)r.ws->place.p;
  b->h_place = r.place;   // the batch's own copy: the upload and the write kernel need not be waited for here
  e = hipMemcpyAsync(d_place, b->h_place.data(), b->h_place.size() * sizeof(LoadPlace), hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = launch_load_write(r.d_raw, r.d_files, d_place, r.n_bufs, (float*)b->d_pcm, b->stream);
  return e == hipSuccess ? AFX_OK : hip_fail(e, "load_write");
}

}  // namespace

}  // namespace host
}  // namespace afx

using namespace afx::host;

extern "C" {

int afx_batch_create(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask,
                     afx_batch** out_batch) {
  if (!plan || !out_batch || n_bufs < 0 || (n_bufs > 0 && !bufs))
    return fail(AFX_ERR_INVALID_ARG, "null argument");
  return batch_create_typed(plan, bufs, n_bufs, mask, first_valid_dtype(bufs, n_bufs), out_batch);
}

// LoadSample front end (SampleAnalyser.cpp:484-718) on the GPU: decoded interleaved PCM in,
// peak-normalised, silence-trimmed, padded mono signal in the analysis arena out.
int afx_batch_create_from_raw(afx_plan* plan, const afx_raw* raws, int32_t n_bufs, uint32_t mask,
                              afx_batch** out_batch, afx_load_info* info) {
  if (!plan || !out_batch || n_bufs < 0 || (n_bufs > 0 && !raws))
    return fail(AFX_ERR_INVALID_ARG, "null argument");
  *out_batch = nullptr;
  if (!mask_ok(mask)) return fail(AFX_ERR_INVALID_ARG, "bad descriptor mask");
  HIP_TRY(hipSetDevice(plan->desc.device));

  RawBatch r{plan, raws, n_bufs};
  r.status.assign((size_t)n_bufs, AFX_OK);
  r.files.resize((size_t)n_bufs);
  r.file_rate.assign((size_t)n_bufs, 0);
  r.scan.resize((size_t)n_bufs);
  r.place.resize((size_t)n_bufs);
  r.lengths.assign((size_t)n_bufs, 0);
  lay_out_raw_arena(r);
  plan_conversions(r);

  const long long t_begin = now_ns();
  // device staging of the decoded PCM and the scan results lives in the batch's pooled workspace
  hipError_t e = hipSuccess;
  r.ws = ws_acquire(plan, &e);
  if (!r.ws) return hip_fail(e, "workspace");
  if (n_bufs > 0) {
    const char* what = "";
    if ((e = stage_and_scan(r, &what)) != hipSuccess) {
      ws_release(plan, r.ws);
      return hip_fail(e, what);
    }
  }
  const long long t_scanned = now_ns();
  place_files(r, info);
  // the workspace (with the staged PCM in it) moves into the batch; build_batch releases it on failure.
  // The arena keeps LoadSample's float signal; a sample of TSampleData::mData is (double)float * FinalScaling
  // (SampleAnalyser.cpp:710-718), formed by the kernels as they load it
  std::vector<int64_t> file_samples((size_t)n_bufs, 0);
  std::vector<int32_t> file_offset((size_t)n_bufs, 0);
  std::vector<double> scales((size_t)n_bufs, 1.0);
  for (int i = 0; i < n_bufs; ++i)
    if (r.status[(size_t)i] == AFX_OK) {
      const afx::LoadPlace& pl = r.place[(size_t)i];
      scales[(size_t)i] = pl.scaling;
      file_samples[(size_t)i] = raws[i].n_frames;                      // mOriginalNumberOfSamples, SampleAnalyser.cpp:464 (before the conversion)
      file_offset[(size_t)i] = (int32_t)(-pl.lead + pl.start_pad);     // mDataOffset, SampleAnalyser.cpp:701
    }
  const long long t_placed = now_ns();
  // the decoded PCM has arrived (the scan was waited for): nothing of the caller's is read after this point
  BatchSource src;
  src.n_bufs = n_bufs; src.mask = mask; src.dtype = afx::kPcmScaledF32; src.lengths = &r.lengths; src.status = &r.status;
  src.fill = fill_from_staging; src.fill_ctx = &r; src.acquired = r.ws;
  src.file_samples = &file_samples; src.file_offset = &file_offset; src.file_rate = &r.file_rate; src.scales = &scales;
  src.wait_for_uploads = false;
  const int st = build_batch(plan, src, out_batch);
  if (g_create_timing.on) {
    const long long t_end = now_ns();
    g_create_timing.ns[0] += t_scanned - t_begin; g_create_timing.ns[1] += t_placed - t_scanned;
    g_create_timing.ns[2] += t_end - t_placed; g_create_timing.ns[3] += 1;
  }
  return st;
}

int afx_extract_batch(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask, afx_out* out) {
  if (!out) return fail(AFX_ERR_INVALID_ARG, "null output");
  if (!plan || n_bufs < 0 || (n_bufs > 0 && !bufs)) return fail(AFX_ERR_INVALID_ARG, "null argument");
  // Large calls (a crawler handing over 10^5 files) are cut into groups of buffers of at most
  // kSplitFrames frames so that the device workspace (8 KiB of magnitudes per frame when the band
  // descriptors are on) stays bounded; results land at the right rows of the caller's arrays.
  constexpr int64_t kSplitFrames = 1 << 19;
  int64_t total = 0;
  for (int i = 0; i < n_bufs; ++i) total += (bufs[i].n_samples > 0) ? num_frames(plan, bufs[i].n_samples) : 0;
  if (total <= kSplitFrames || n_bufs <= 1) {
    afx_batch* b = nullptr;
    int st = afx_batch_create(plan, bufs, n_bufs, mask, &b);
    if (st != AFX_OK) return st;
    st = afx_batch_run(b);
    if (st == AFX_OK) st = afx_batch_fetch(b, out);
    afx_batch_destroy(b);
    return st;
  }
  const int call_dtype = first_valid_dtype(bufs, n_bufs);
  struct Col { double* afx_out::*field; int width; };
  std::vector<Col> cols;
  for (const FieldDesc& d : kFields) cols.push_back(Col{d.out, d.width});
  cols.push_back(Col{&afx_out::magnitude, 1024});
  int64_t row0 = 0;
  int32_t first = 0;
  if (out->frame_offset) out->frame_offset[0] = 0;
  while (first < n_bufs) {
    int32_t last = first;
    int64_t group = 0;
    while (last < n_bufs) {
      const int64_t f = (bufs[last].n_samples > 0) ? num_frames(plan, bufs[last].n_samples) : 0;
      if (last > first && group + f > kSplitFrames) break;
      group += f;
      ++last;
    }
    afx_out part = *out;
    for (const Col& c : cols)
      if (out->*(c.field)) part.*(c.field) = out->*(c.field) + row0 * c.width;
    std::vector<int64_t> off((size_t)(last - first) + 1);
    part.frame_offset = off.data();
    part.buf_status = out->buf_status ? out->buf_status + first : nullptr;
    part.effective_length = out->effective_length ? out->effective_length + (size_t)first * 3 : nullptr;
    afx_batch* b = nullptr;
    int st = batch_create_typed(plan, bufs + first, last - first, mask, call_dtype, &b);
    if (st != AFX_OK) return st;
    st = afx_batch_run(b);
    if (st == AFX_OK) st = afx_batch_fetch(b, &part);
    afx_batch_destroy(b);
    if (st != AFX_OK) return st;
    if (out->frame_offset)
      for (int32_t i = first; i < last; ++i) out->frame_offset[i + 1] = row0 + off[(size_t)(i - first) + 1];
    row0 += off.back();
    first = last;
  }
  return AFX_OK;
}

}  // extern "C"
