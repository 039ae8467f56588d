// afec_amd/csrc/afx_internal.h -- shared between the C-ABI host code and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afx {
constexpr uint32_t kFramesWholeSpectrum = 1u << 31;
// internal bit of FrameArgs::mask: bands_kernel runs for this batch anyway and takes the spectral statistics (bits 1..7)
// from the stored magnitudes; the half-wave frame kernel then only stores them (its magnitude class)
constexpr uint32_t kFramesStatsLater = 1u << 30;
// raw sums per frame the half-wave statistics classes (or bands_kernel for them) leave for stats32_finish_kernel: sum m,
// m^2, j m, j^2 m, m^3, m^4, sum log(m + 1e-20), rolloff count (+ two spare slots)
constexpr int kStatTmp = 10;

// Geometry the kernels are specialised for: the only one the reference ever instantiates
// (Crawler.cpp:41-43, 599-600): 44.1 kHz, 2048-sample frame, 1024 hop.
constexpr int kSampleRate = 44100;
constexpr int kFft = 2048;
constexpr int kHop = 1024;
constexpr int kHalf = kFft / 2;   // magnitude bins used (SampleAnalyser.cpp:837-838)
constexpr int kNumCep = 14;       // SampleDescriptors.h:464
constexpr int kNumBands = 28;
constexpr int kNumSub = 14;
// SampleAnalyser.cpp:171-175 with integer division 44100/2048 = 21
constexpr int kFirstBin = 1;
constexpr int kLastBin = 738;
constexpr int kBinCount = 738;

// mel filter support on the 1024-bin grid (LibXtract init.c:237-382 with M = N>>1, measured from
// the table the plan builds; afx_plan_create re-verifies the runtime table against this cover)
constexpr int kMelLo[kNumCep] = {1, 1, 5, 10, 17, 25, 35, 48, 64, 83, 108, 139, 177, 225};
constexpr int kMelHi[kNumCep] = {3, 8, 15, 23, 33, 46, 62, 81, 106, 137, 175, 223, 283, 358};
constexpr int kMelRows = 6;       // bins 0..383 = rows r = 0..5 of the [r][lane] layout

// does mel filter f touch row r (bins 64r .. 64r+63)?
constexpr bool mel_touches(int f, int r) { return kMelLo[f] <= 64 * r + 63 && kMelHi[f] >= 64 * r; }
constexpr int mel_pair_count() {
  int n = 0;
  for (int r = 0; r < kMelRows; ++r)
    for (int f = 0; f < kNumCep; ++f) n += mel_touches(f, r) ? 1 : 0;
  return n;
}
constexpr int kMelPairs = mel_pair_count();  // 22
// index of pair (r, f) in the packed weight table
constexpr int mel_pair_index(int r, int f) {
  int n = 0;
  for (int rr = 0; rr < kMelRows; ++rr)
    for (int ff = 0; ff < kNumCep; ++ff) {
      if (rr == r && ff == f) return n;
      n += mel_touches(ff, rr) ? 1 : 0;
    }
  return -1;
}

// the same cover on rows of 32 bins (half-wave layout of afx_frames32.hip: bin = q + 32 r)
constexpr int kMel32Rows = 12;    // bins 0..383
constexpr bool mel32_touches(int f, int r) { return kMelLo[f] <= 32 * r + 31 && kMelHi[f] >= 32 * r; }
constexpr int mel32_pair_count() {
  int n = 0;
  for (int r = 0; r < kMel32Rows; ++r)
    for (int f = 0; f < kNumCep; ++f) n += mel32_touches(f, r) ? 1 : 0;
  return n;
}
constexpr int kMel32Pairs = mel32_pair_count();  // 31
constexpr int mel32_pair_index(int r, int f) {
  int n = 0;
  for (int rr = 0; rr < kMel32Rows; ++rr)
    for (int ff = 0; ff < kNumCep; ++ff) {
      if (rr == r && ff == f) return n;
      n += mel32_touches(ff, rr) ? 1 : 0;
    }
  return -1;
}

// 28 "frequency_bands" edges in bins: round(f/21) of SampleAnalyser.cpp:2015-2019, first start =
// bin 1, last end clamped to 1024 (SampleAnalyser.cpp:2029-2040)
constexpr int kBandEdge[kNumBands + 1] = {1, 2, 5, 7, 10, 14, 19, 24, 30, 37, 44, 51, 60, 70, 82,
                                          95, 110, 129, 150, 176, 210, 252, 305, 367, 452, 571, 738,
                                          905, 1024};
constexpr bool band_touches(int b, int r) {
  return kBandEdge[b] <= 64 * r + 63 && kBandEdge[b + 1] - 1 >= 64 * r;
}

// 14 sub-bands of CalcSpectralBandFeatures: bin counts EndBin - StartBin + 1 of the nominal edges
// 50..15500 Hz at 21 Hz/bin (SampleAnalyser.cpp:2077-2100), laid out contiguously from bin 1
constexpr int kSubN[kNumSub] = {2, 4, 6, 10, 12, 15, 17, 23, 29, 41, 61, 96, 148, 287};

// PCM element types of the analysis arena.  kPcmF32 / kPcmF64 are the ABI's AFX_PCM_F32 / AFX_PCM_F64 (buffers the
// caller normalised itself).  kPcmScaledF32 is what the LoadSample front end leaves: the float mono signal of the file
// (TSampleAnalyser::LoadSample's AnalyzationSampleBuffer, a TArray<float>, SampleAnalyser.cpp:487-491, 534-561, 625-631)
// plus one double per buffer, FinalScaling -- the reference's TSampleData::mData[n] is exactly
// (double)AnalyzationSampleBuffer[n + lead] * FinalScaling (SampleAnalyser.cpp:710-718), which every consumer forms
// when it loads a sample: the same doubles at half the bytes.
enum { kPcmF32 = 0, kPcmF64 = 1, kPcmScaledF32 = 2 };

// chunk = a run of consecutive frames of one buffer processed by one wave
struct Chunk {
  int64_t sample_off;  // offset of the first processed frame's first sample in the PCM arena
  int32_t frame0;      // global output row of the first *emitted* frame
  int16_t nframes;     // emitted frames
  int16_t flags;
  double scale;        // kPcmScaledF32: the buffer's FinalScaling (1.0 otherwise)
};
// companion of Chunk for the time-domain kernels: samples of the buffer from the chunk's first frame on
// (TSampleData::mData.Size() - n, SampleAnalyser.cpp:943), saturated at 2^30
using ChunkRemaining = int32_t;
enum { kChunkFirstOfBuffer = 1 };

// per-frame output record: offsets in doubles, -1 = not selected
struct RecordLayout {
  int32_t stride;
  int32_t mfcc, srms, centroid, spread, skew, kurt, rolloff, flatness, flux, bands, amp_peak,
      amp_rms, sub_rms, sub_flat, sub_flux, sub_cplx, sub_contrast, contrast;
  // neighbours
  int32_t silence, envelope, complexity, autocorr, f0, f0_conf, f0_safe, inharm, tri1, tri2, tri3;
};

struct FrameArgs {
  const void* pcm;
  const Chunk* chunks;
  int32_t n_chunks;
  uint32_t mask;
  double* rec;
  RecordLayout lay;
  double* mag_out;   // [F][1024] or nullptr
  const void* win;   // [16][64] pairs (w[2n], w[2n+1]) / 4096, n = 64 r + lane      (T)
  const void* t1;    // [4][4][4] complex w64^(m2*(4*jh+jl)), index 16*jh + 4*m2 + jl        (T)
  const void* t2;    // [16][64] complex w1024^(n2*(4*jh+jl+16*j2)), reg=4*j2+jl, lane=16*jh+n2 (T)
  const void* post;  // [16][64] complex w2048^(lane+64r)                             (T)
  const void* melw;    // [kMelPairs][64] packed mel rows                                  (T)
  const double* dct;   // [14][16]: cos(pi * (n/14) * (m + 0.5)), m < 14
  // tables of the half-wave kernels (afx_frames32.hip), always double
  const void* win32;   // [32][32] pairs (w[2n], w[2n+1]) / 4096, n = q + 32 n1
  const void* tw32;    // [32][32] complex w1024^(n2 k1), index 32 n2 + k1
  const void* post32;  // [32][32] complex w2048^(q + 32 r)
  const void* melw32;  // [kMel32Pairs][32] packed mel rows
  unsigned* queue;     // work-queue counter of the half-wave kernels: advances by ceil(n_chunks / 2) per launch
  unsigned queue_base; // its value when this launch starts
  unsigned long long* stamps;   // diagnostic builds (AFX_STAMPS): 16 per-stage cycle counters, else nullptr
  double* stat_tmp;             // half-wave statistics / full class: [F][frames32_stat_tmp_doubles()] raw sums per frame (afx_frames32.hip)
  int64_t mag_spare_row;        // half-wave full class: row of mag_out behind the last frame (stores of frames past a chunk's end)
};

// launchers (afx_kernels.hip).  precision: 0 = f64, 1 = f32; pcm_dtype: kPcm*
hipError_t launch_frames(const FrameArgs& a, int precision, int pcm_dtype, int grid_blocks,
                         hipStream_t stream);
int frames_waves_per_block(uint32_t mask);    // waves (of 64 lanes) per workgroup
// half-wave kernels (afx_frames32.hip): a wave walks two chunks at a time, one per 32-lane half
bool frames_use_halfwave(uint32_t mask, int precision, int pcm_dtype);
hipError_t launch_frames32(const FrameArgs& a, int grid_blocks, hipStream_t stream, int64_t total_frames, bool scaled);
// closed forms of the spectral statistics from the raw sums in FrameArgs::stat_tmp (after bands_kernel with kBandsStats)
hipError_t launch_stats32_finish(const FrameArgs& a, hipStream_t stream, int64_t total_frames);
int frames32_waves_per_block();
int frames32_stat_tmp_doubles();
int frames32_class(uint32_t frames_mask);   // 0 = MFCC only, 1 = + spectral statistics, 2 = full (bins 0..768 stored, amplitude), 3 = full, whole spectrum
// internal bit of FrameArgs::mask: a consumer of the stored magnitudes needs bins above 768 too (the whitening follower
// and the fail-safe f0 of afx_whiten.hip, the magnitude output) -- the half-wave full class then produces them
int frames_feature_class(uint32_t mask);     // 0 = MFCC only, 1 = + statistics, 2 = everything

// Work queue of a persistent-grid launch: a wave starts with the item of its own index and draws every further one from
// a device counter (one atomic per item), instead of a static stride -- of the two waves that share a SIMD the older one
// wins the issue arbitration and CUs differ, so equal shares do not finish together.  The counter is never reset: a
// launch advances it by exactly its number of items (a draw follows every processed item), and the host passes its value
// at launch.  counter == nullptr: static stride.
struct WorkQueue {
  unsigned* counter;
  unsigned base;
};
enum { kQueueFrames32 = 0, kQueuePitch = 1, kQueueAcorr = 2, kQueueBands = 3, kQueueWhiten = 4, kQueueSlots = 16 };
// workgroups of `kernel` that are resident on the device at once (a WorkQueue launch must not be larger: the static first
// items of workgroups that start late would be processed last, by few waves)
template <typename K>
int resident_blocks(K kernel, int threads, size_t dynamic_lds) {
  int per_cu = 0, cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dynamic_lds) != hipSuccess || per_cu < 1) per_cu = 1;
  return per_cu * cus;
}

// band-feature kernel (SampleAnalyser.cpp:2067-2308) working from stored magnitudes
struct BandArgs {
  WorkQueue queue;
  const double* mag;    // [F][1024]
  const Chunk* chunks;  // runs of consecutive frames of one buffer (kChunkFirstOfBuffer marks frame 0)
  int32_t n_chunks;
  double* rec;          // same per-frame records the frame kernel writes
  RecordLayout lay;
  uint32_t flags;       // kBands*
  double* stat_tmp;     // kBandsStats: [F][frames32_stat_tmp_doubles()] raw sums per frame (as the half-wave statistics class leaves them)
};
// kBandsSpectrum: the 28 "frequency_bands" (SA:2007-2048); kBandsStats: the raw sums of the spectral statistics over bins
// 1..738 (SA:1808-1915) into BandArgs::stat_tmp, for stats32_finish_kernel
enum { kBandsFeatures = 1, kBandsFlux = 2, kBandsSpectrum = 4, kBandsStats = 8 };
hipError_t launch_bands(const BandArgs& a, hipStream_t stream);

// ---- neighbours of the spectral set ----
// time-domain descriptors of every frame, one wave per chunk (afx_time.hip)
struct TimeArgs {
  WorkQueue queue;        // pitch_kernel, acorr_kernel (set per launch)
  const void* pcm;
  const Chunk* chunks;
  const ChunkRemaining* remaining;
  int32_t n_chunks;
  int32_t pcm_dtype;      // kPcm*
  double* rec;
  RecordLayout lay;
  const void* t1;         // the double FFT tables of FrameArgs
  const void* t2;
  const void* post;
  uint32_t amplitude;     // hop_kernel, for the half-wave full classes: AFX_D_AMPLITUDE_PEAK | AFX_D_AMPLITUDE_RMS to write
  uint32_t hop_here;      // pitch_kernel also writes the hop's descriptors (silence, envelope, amplitude): no hop_kernel launch
};
hipError_t launch_hop(const TimeArgs& a, hipStream_t stream);      // silence flag, envelope (+ amplitude peak / rms)
hipError_t launch_acorr(const TimeArgs& a, hipStream_t stream);    // auto_correlation
hipError_t launch_pitch(const TimeArgs& a, hipStream_t stream);    // f0, f0 confidence (aubio yinfast)

// whitening follower over the frames of each buffer + peak spectrum + fail-safe f0 (afx_whiten.hip), from
// the stored magnitudes; runs after launch_pitch.  Two kernels: the follower recurrence alone, one wave per
// (buffer, 64 bins), leaves its state at every chunk start; then one wave per chunk does the per-frame work.
struct WhitenArgs {
  WorkQueue queue;              // whiten_kernel
  const double* mag;            // [F][1024]
  const int64_t* frame_offset;  // [n_bufs + 1], device
  const int32_t* chunk_first;   // [n_bufs + 1]: index of the buffer's first chunk in `chunks`
  const Chunk* chunks;          // the whitening kernels' own chunk table (afx_batch_plan.cpp, cut_whitening_chunks)
  int32_t n_bufs, n_chunks;
  int32_t chunk_frames;         // frames per chunk of the buffers that have several (their last chunk may be shorter)
  int32_t need_follow;          // some buffer has more than one chunk: follow_kernel leaves the state at their starts
  uint32_t mask;                // AFX_D_* bits
  double* rec;
  RecordLayout lay;
  double* follower;             // [n_chunks][1024] follower state at the chunk's first frame
  double decay, floor_value;    // aubio_spectral_whitening: r_decay, floor
};
hipError_t launch_whiten(const WhitenArgs& a, hipStream_t stream);

// effective length (SampleAnalyser.cpp:1715-1755): silent leading / trailing samples of every buffer at
// three floors; out[n_bufs][6] = first, last sample above the floor (INT_MAX, -1 when none) at -48, -24, -12 dB
struct BufSpan {
  int64_t off;   // first sample of the buffer in the PCM arena
  int64_t n;     // samples (the whole buffer)
  double scale;  // kPcmScaledF32: the buffer's FinalScaling (1.0 otherwise)
};
hipError_t launch_effective_length(const void* pcm, int pcm_dtype, const BufSpan* spans, int n_bufs, double floor48,
                                   double floor24, double floor12, int32_t* out, hipStream_t stream);

// per-buffer statistics of every record column (TStatistics::Calc, Statistics.cpp:12-90)
struct StatsArgs {
  const double* rec;            // [F][stride]
  const int64_t* frame_offset;  // [n_bufs + 1], device
  int32_t n_bufs;
  int32_t stride;
  double* stats;                // [n_bufs][stride][13]
  int32_t small_rows;           // longest series of 2..128 frames in the batch (0: none): lane-per-series kernel
  int32_t need_long;            // some buffer has < 2 or > 128 frames: wave-per-series kernel
};
hipError_t launch_stats(const StatsArgs& a, hipStream_t stream);

// ---- rhythm tracker (SampleAnalyser.cpp:983-1048; afx_rhythm.hip) ----
struct RhythmFile {
  int64_t sample_off;   // first sample of the buffer in the PCM arena
  int64_t frame0;       // first row of the buffer's 512/128 frames in the batch-wide arrays
  int32_t frames;       // 512/128 frames of the analysed prefix (SampleAnalyser.cpp:991)
  int32_t long_slot;    // > 0: the file's onset functions come from the long-file kernels (slot long_slot - 1)
  double scale;         // kPcmScaledF32: the buffer's FinalScaling (1.0 otherwise)
  double duration_s;    // SampleDurationInSeconds (SampleAnalyser.cpp:1001-1002)
  double offset_s;      // OnsetOffsetInSeconds    (SampleAnalyser.cpp:1003-1004)
};
constexpr int kRhythmLdsFrames = 8192;   // longest onset series the post kernel stages in LDS (64 KiB; 20 s = 6887 frames)
constexpr int kRayleighTable = 8192;   // beyond it the Rayleigh weight of the beat tracker has underflowed to 0
struct RhythmArgs {
  const void* pcm;
  int32_t pcm_dtype;            // kPcm*
  int32_t n_files;
  const RhythmFile* files;
  int64_t total_frames;
  int32_t sample_rate;
  int32_t rayleigh_n;
  const double* window;         // [512]: 0.5 x Hanning (the 1/2 of the real-input untangle folded in)
  const double* tw256;          // [16][16] complex: w256^(n2 k1) at [n2][k1]
  const double* ut512;          // [16][16] complex: w512^(q + 16 r) at [r][q]
  const double* canny;          // [25]: TCannyWindow(12, 16) coefficients
  const double* rayleigh;       // [rayleigh_n]: rwv of aubio's beat tracker
  float relax_coef, norm_complex, norm_power;
  float thresh[2];              // complex, percussive
  int32_t medspan;
  int32_t mingap[2];
  int32_t lds_frames;           // frames of the longest file that fits kRhythmLdsFrames (0: none)
  float* odf;                   // [total_frames][2]: onset functions before median removal
  double* onsets;               // [total_frames][2]: TRhythmTracker::Onsets (complex, percussive)
  double* scratch;              // [8][total_frames]
  double* scalars;              // [n_files][14]
  // long files of a small batch: onset functions by three kernels instead of one workgroup per file (afx_rhythm.hip)
  int32_t n_long, long_rounds;          // files on that path; rounds of 16 frames they have together
  const int32_t* long_files;            // [n_long]: index into files
  const int32_t* long_round_off;        // [n_long + 1]: first round of each
  const int64_t* long_frame_off;        // [n_long + 1]: first row of each in long_polar
  float2* long_polar;                   // [long frames][256]: (magnitude, phase)
  float* long_den;                      // [long frames][256]: what the whitening divides the magnitude by
};
// which files take the long-file path: a batch of at most this many files (one workgroup per file leaves CUs idle and
// runs hundreds of dependent rounds; measured on 20 s files: 1 file 5.8 -> 0.35 ms, 256 files 5.9 -> 5.8 ms: the
// paths meet there) and a file of at least this many 512/128 frames (3 s)
constexpr int kRhythmLongBatchFiles = 256, kRhythmLongFrames = 1024;
// rows of a long file in long_polar / long_den: its frames rounded up to a multiple of this (the follower kernel's batches)
constexpr int kRhythmLongPad = 48;
hipError_t launch_rhythm(const RhythmArgs& a, hipStream_t stream);

// ---- LoadSample front end (SampleAnalyser.cpp:484-718) ----
struct LoadFile {
  int64_t raw_off;   // byte offset of the interleaved PCM in the raw arena
  int64_t n_frames;  // sample frames
  int32_t channels;
  int32_t format;    // AFX_RAW_*
};
struct LoadScan {    // per file, produced by the scan kernels
  double sum_sq;         // sum (x / 32768)^2 of the mono mix
  double amplification;  // 32768 / max|x|  (1 when silent)
  float max_amp;         // max |x| of the mono mix in "16-bit float" units
  int32_t lead, trail;   // first / last sample above -48 dB after normalisation (INT_MAX / -1 when none)
  int32_t pad;
};
struct LoadPlace {   // per file, where the normalised samples go
  int64_t out_off;   // first sample of the buffer in the analysis arena
  int64_t out_n;     // samples of the analysed prefix kept in the arena
  int64_t lead;      // first audible source sample
  int64_t audible;   // audible samples
  int64_t start_pad; // zeros in front (StartFrameOffset)
  double scaling;    // FinalScaling
};
constexpr int kRawMonoFloat = 5;   // LoadFile::format of a converted file: mono floats already in the "16-bit float" range
int load_scan_blocks_per_file(int n_files);      // partial_scratch holds n_files x this x 16 bytes
hipError_t launch_load_scan(const unsigned char* raw, const LoadFile* files, int n_files, double silence_floor,
                            void* partial_scratch, LoadScan* scan, hipStream_t stream);
hipError_t launch_load_write(const unsigned char* raw, const LoadFile* files, const LoadPlace* place, int n_files,
                             float* arena, hipStream_t stream);

// ---- sample-rate conversion in front of the LoadSample kernels (SampleAnalyser.cpp:563-607 -> libresample; afx_resample.hip) ----
// zero samples on either side of a file's mono mix: the filter's reach at the smallest supported factor (1/16: 18 x 16
// input samples a wing, + the converter's creep)
constexpr int kResampleMargin = 320;
struct ResampleFile {
  int64_t raw_off;    // the file's decoded PCM in the raw arena (bytes)
  int64_t mono_off;   // its mono mix, kResampleMargin + n_in + kResampleMargin floats (bytes from the arena's start, 16-byte aligned)
  int64_t out_off;    // the converted samples, n_out floats
  int64_t group_off;  // the file's first record (one per 16 output samples)
  int64_t block_off;  // the file's first workgroup of the filter kernel
  int64_t n_in, n_out;
  int32_t channels, format;
  double factor;      // analyser rate / file rate = 1 / Speed
};
struct ResampleGroup {   // output samples 16 r .. 16 r + 15 of a file
  double t0, t1;         // the converter's time (relative to its input window) at the first sample; at sample `cut`
  int32_t base0, base1;  // input sample the window starts at, for samples < cut and >= cut
  int32_t cut;           // 16: one window; 1..15: a new window starts at this sample
  int32_t pad;           // 0: all 16 produced; 1: none; 2 + k: the first k (the converter ended early, SA:596-597)
};
int64_t resample_blocks(int64_t n_out);   // workgroups of the filter kernel for a file of n_out converted samples
hipError_t launch_resample(unsigned char* raw, const ResampleFile* files, int n_files, int64_t n_blocks, int64_t max_n_in,
                           ResampleGroup* groups, const float* filter, hipStream_t stream);

}  // namespace afx
