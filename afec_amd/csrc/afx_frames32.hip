// afec_amd/csrc/afx_frames32.hip -- gfx950 STFT + MFCC kernel, one frame per HALF-wave.
//
// A wavefront is two independent 32-lane halves; each half owns a run of consecutive frames of one buffer
// ("chunk") and computes the 2048-point FFT of the windowed real frame as a 1024-point complex FFT of
// z[n] = x[2n] + i x[2n+1] held as 32 complex doubles per lane (lane = 32 h + q):
//
//   load   v[n1] = z[q + 32 n1]                      (hop reuse: rows 0..15 are last frame's 16..31, kept in
//          registers; the new hop arrives by LDS-DMA in the wave's exchange plane, issued one frame ahead)
//   P1     32-point DFT over n1, in registers        -> register k1
//   E      exchange through the wave's LDS plane, slot = 1056 h + 33 k1 + n2 (conflict-free for
//          ds_write_b64 and ds_read_b64)             -> lane q = k1, register n2
//   T      * w1024^(n2 k1)                           (LDS table [n2][k1])
//   P2     32-point DFT over n2, in registers        -> v[k2] = Z[q + 32 k2]
//   U      partner Z[1024-k] by a cross-lane read inside the half, even/odd untangle, |X[k]|, k = q + 32 r
//
// then the descriptors straight from the magnitudes in registers; sums over a frame's bins are reductions over
// the 32 lanes of a half (v_permlane16_swap + DPP), so one instruction stream serves two frames.  Compared with
// the 64-lane layout of afx_kernels.hip (16 x 4 x 16, two exchanges) this drops the register-transpose exchange
// (64 v_permlane*_swap per frame), one of the two twiddle stages and half of the reduction work.  Index algebra,
// LDS bank rules and the reduction's lane map are modelled in tools/fft32_dataflow_model.py.
//
// What each stage replaces in the reference (SampleAnalyser.cpp = SA):
//   window+FFT+magnitude  SA:826-845 (xtract_windowed, TFftTransformComplex, TAudioMath::Magnitude)
//   mel+log+DCT           SA:2052-2063 -> LibXtract vector.c:350-391

#include <hip/hip_runtime.h>

#include "afx_internal.h"
#include "afx_device.h"
#include "afx_fft32.h"

namespace afx {
namespace {

using f32x32::cx;

constexpr int kHalfSlots = 1056;                      // 8-byte slots per half: 33 * 31 + 31 + 1, rounded to 32
constexpr int kPlane32Bytes = 2 * kHalfSlots * 8;     // 16 896 per wave
constexpr int kWaves32 = 8;   // (six waves with 22.5 KB of LDS each were measured: 5-10 % slower, HISTORY.md round 5)
// raw sums per frame the statistics / full classes leave for stats32_finish_kernel: sum m, m^2, j m, j^2 m, m^3, m^4,
// sum log(m + 1e-20), rolloff count, sum x^2 of the hop, max |x| of the hop

template <int POST_ROWS>
struct Lds32 {
  static constexpr int tw = 0;                              // [32][32] cx<double>: w1024^(n2 k1)
  static constexpr int post = tw + 32 * 32 * 16;            // [POST_ROWS][32] cx<double>: w2048^(q + 32 r)
  static constexpr int dct = post + POST_ROWS * 32 * 16;    // [14][16] double
  static constexpr int logc = dct + 14 * 16 * 8;            // 32 doubles: constants of the logarithm and of the statistics (kLogConst)
  static constexpr int xchg = logc + 32 * 8;                // kWaves32 planes
  static constexpr int total = xchg + kWaves32 * kPlane32Bytes;
};

// ---- constant tables in global memory through buffer descriptors ----
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 table_load2(__amdgpu_buffer_rsrc_t rs, int lane_off, int row_off) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, row_off, 0));
}
// row offset as the scalar offset: for a lane offset the compiler cannot see through
__device__ __forceinline__ double table_load1(__amdgpu_buffer_rsrc_t rs, int lane_off, int row_off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, row_off, 0));
}

// LDS-DMA of one 1 KiB piece (16 bytes per lane: lane L's bytes land at lds_base + IMM + 16 L, read from
// gaddr + IMM; non-temporal: the PCM is read once).  Inline assembly on purpose: for the builtin the compiler's
// wait-count pass assumes that every later DS read may alias the DMA's LDS write and puts s_waitcnt vmcnt(0) in
// front of it.  The kernel orders the DMA against its own DS traffic by hand: it is issued behind an explicit
// s_waitcnt lgkmcnt(0) (the exchange reads have returned) and the hop is read behind an explicit s_waitcnt vmcnt(0).
// M0 has no other user in this kernel (gfx9 DS instructions do not read it).
// NT: non-temporal (the classes that read the PCM once); the classes that fetch a hop a second time one frame later, as
// the next frame's overlap half, leave the first read to the cache's normal policy.
template <int IMM, bool NT>
__device__ __forceinline__ void dma_piece(const void* gaddr, unsigned lds_base) {
  if constexpr (NT)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2 nt"
                 :: "v"(gaddr), "s"(lds_base), "n"(IMM) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                 :: "v"(gaddr), "s"(lds_base), "n"(IMM) : "memory", "m0");
}
// the eight pieces of a wave's hop image: global piece j at +512 j bytes, LDS piece j at +1024 j
template <bool NT = true>
__device__ __forceinline__ void dma_hop(const float* g, unsigned lds_plane) {
  dma_piece<0, NT>(g, lds_plane);
  dma_piece<512, NT>(g, lds_plane + 512);
  dma_piece<1024, NT>(g, lds_plane + 1024);
  dma_piece<1536, NT>(g, lds_plane + 1536);
  dma_piece<2048, NT>(g, lds_plane + 2048);
  dma_piece<2560, NT>(g, lds_plane + 2560);
  dma_piece<3072, NT>(g, lds_plane + 3072);
  dma_piece<3584, NT>(g, lds_plane + 3584);
}

// Magnitude 2 sqrt(xr^2 + xi^2 + DBL_MIN): v_rsq_f64 seed (~2^-23) + one Newton step y (3 - r y) (~2e-14 relative)
// without its division by two -- the window table carries that 1/2, so the arguments are the halved spectrum and the
// result is |X|.  The DBL_MIN keeps the seed finite for an all-zero bin; sums of squares of the true spectrum below
// DBL_MIN -- which the reference flushes to 0 (TAudioMath::Magnitude runs with DAZ + FZ, AudioMath.cpp:25-35) -- come
// out as at most sqrt(5 DBL_MIN) = kFlushLevel: every mel sum of such bins stays below the 2e-42 floor of the
// logarithm, and the statistics class drops them by comparing with that level.  tiny / three: SGPR pairs (a 64-bit
// literal would cost a VGPR pair for the whole kernel).
__device__ __forceinline__ double mag_sqrt_mel(double xr, double xi, double tiny, double three) {
  const double x = fma(xi, xi, fma(xr, xr, tiny));
  const double r = __builtin_amdgcn_rsq(x);
  const double y = x * r;
  return y * fma(-r, y, three);
}

// Lanes 0 and 32 (bins 32 r) are their own partners, at another register than everybody else's: two 64-bit moves
// under EXEC = {0, 32} put their values in place, instead of four v_cndmask_b32 per row.  EXEC is saved and restored
// (the kernel's control flow is wave-uniform where this is called, but nothing here depends on that).
__device__ __forceinline__ void keep_lane0(double& re, double& im, double own_re, double own_im, unsigned long long lane0_mask) {
  unsigned long long saved;
  asm volatile("s_mov_b64 %2, exec\n\ts_mov_b64 exec, %5\n\tv_mov_b64 %0, %3\n\tv_mov_b64 %1, %4\n\ts_mov_b64 exec, %2"
               : "+v"(re), "+v"(im), "=&s"(saved) : "v"(own_re), "v"(own_im), "s"(lane0_mask));
}

// ---- reduction of 16 per-lane values over the 32 lanes of each half, in registers: lane L ends with the total of
// a[(L & 31) >> 1] ----
__device__ __forceinline__ double half_sum16(double (&a)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    swap16(a[i], a[i + 8]);
    a[i] += a[i + 8];
  }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double keep = b3 ? a[i + 4] : a[i];
    const double send = b3 ? a[i] : a[i + 4];
    a[i] = keep + dpp_mov<kDppRor8>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double keep = b2 ? a[i + 2] : a[i];
    const double send = b2 ? a[i] : a[i + 2];
    a[i] = keep + dpp_mov<kDppHalfMirror>(send);
  }
  const double keep = b1 ? a[1] : a[0];
  const double send = b1 ? a[0] : a[1];
  double z = keep + dpp_mov<kDppXor2>(send);
  z += dpp_mov<kDppXor1>(z);
  return z;
}

// Constants of finish_mfcc32's logarithm.  They live in LDS and are read where they are used: as literals the
// compiler keeps every one of them in a vector register pair for the whole kernel (no 64-bit literals on
// gfx950's VOP3), which this kernel cannot afford.
__device__ const double kLogConst[32] = {1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0,
                                         1.0 / 11.0, 1.0 / 9.0,  1.0 / 7.0,  1.0 / 5.0,  1.0 / 3.0,  0.693147180559945309417,
                                         0.70710678118654752440, 2e-42,
                                         // constants of the statistics class (kC* below)
                                         1e-20, 32.0, 0.38268343236508984, -0.9238795325112867, 1.0 / 738.0,
                                         (double)(85.0f / 100.0), 43.0, -1.0 / 60.0, (double)1e-12f, 3.0,
                                         1.44269504088896340736, 20.0 / 2.30258509299404568402, 2.2250738585072014e-308, 3.33547137486383e-154 /* kFlushLevel = sqrt(5 DBL_MIN) */, 511.0, 0.0, 0.0, 0.0};
// indices into kLogConst of the statistics class' constants
constexpr int kCEps = 14, kC32 = 15, kCRotRe = 16, kCRotIm = 17, kCRoll = 19, kCSqrtMin = 27, kC511 = 28;
// (entries 18, 20..26 -- 1/738, 43, -1/60, 1e-12f, 3, log2 e, 20/ln 10, DBL_MIN -- served the closed forms while they
// lived in this kernel; they are stats32_finish_kernel's literals now)

// natural log of a positive normal double: frexp + atanh series (|s| <= 0.1716), ~1e-16 absolute on log(m)
__device__ __forceinline__ double log_lds(double x, const double* c) {
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lowhalf = m < c[12];
  m = lowhalf ? m + m : m;
  e = lowhalf ? e - 1 : e;
  const double s = fast_div(m - 1.0, m + 1.0);
  const double z = s * s;
  double p = c[0];
#pragma unroll
  for (int i = 1; i < 11; ++i) p = fma(p, z, c[i]);
  p = fma(p, z, 1.0);
  return fma((double)e, c[11], 2.0 * s * p);
}

// log + 14-point DCT-II + store (vector.c:364-391) for the frames whose mel sums sit in lanes 2 f + slot of each
// half.  recp: this lane's output element of slot s = lane & 1 (coefficient n = (lane >> 1) & 15); left: frames
// of this half still to be stored, counted from slot s.
__device__ __forceinline__ void finish_mfcc32(double acc, double*& recp, int& left, int stride2, const double* dct,
                                              const double* logc, int lane) {
  const double lg = log_lds(fmax(acc, logc[13]), logc);  // XTRACT_LOG_LIMIT 2e-42, vector.c:364
  // lane-derived addresses are recomputed here (laundered lane id): hoisted out of the frame loop they would each
  // hold a register for the whole kernel
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int n = (ln >> 1) & 15;
  const double2* drow = reinterpret_cast<const double2*>(dct + 16 * (n < kNumCep ? n : 0));
  const int idx = ((ln & 32) + (ln & 1)) << 2;   // byte index of lane 32 h + s for ds_bpermute
  const int lo32 = __double2loint(lg), hi32 = __double2hiint(lg);
  double c = 0.0;
#pragma unroll
  for (int m = 0; m < kNumCep; m += 2) {
    const double2 d = drow[m >> 1];
    const double l0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(idx + 8 * m, hi32), __builtin_amdgcn_ds_bpermute(idx + 8 * m, lo32));
    const double l1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(idx + 8 * m + 8, hi32), __builtin_amdgcn_ds_bpermute(idx + 8 * m + 8, lo32));
    c = fma(l0, d.x, c);
    c = fma(l1, d.y, c);
  }
  if (n < kNumCep && left > 0) *recp = c;
  recp += stride2;
  left -= 2;
}

// Diagnostic build only (make stamps -> libafx_hip_stamps.so, never the shipped library): wave 0 of every
// workgroup stamps s_memtime at the stage boundaries and adds the differences to per-stage counters.
#ifndef AFX_STAMPS
#define AFX_STAMPS 0
#endif
#if AFX_STAMPS
// scalar-only: s_memtime into an SGPR pair, the per-stage sums stay in SGPRs (no vector registers, no memory traffic)
#define AFX_STAMP(i)                                                                           \
  do {                                                                                         \
    unsigned long long t_;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                           \
    stamp_acc[i] += (unsigned)(t_ - stamp_prev);                                               \
    stamp_prev = t_;                                                                           \
  } while (0)
#else
#define AFX_STAMP(i)
#endif

// Statistics class: untangle the bins in PAIRS.  X[k] and X[1024 - k] come from the same (Z[k], Z[1024 - k]):
// X[k] = E + w O, |X[1024 - k]| = |E - w O|, so row r (bins 32 r + q) also yields the "mirrored block" M_r = bins
// 32 (31 - r) + 1 .. 32 (32 - r) -- in reversed lane order (lane q holds bin 1024 - 32 r - q; lane 0 the block's last
// bin) -- for ten more operations instead of a second fetch + untangle.  Rows 0..15, bin 512 (lane 0 of row 16, its own
// partner) and M_15..M_8 cover the analysis range 1..738: 16 untangle steps instead of 24.

template <int FEAT, bool SCALED>
__global__ __launch_bounds__(kWaves32 * 64) void frames32_kernel(const FrameArgs a) {
  // FEAT 4 ("magnitude" class, for the same masks when bands_kernel runs anyway: flux, spectrum bands or sub-band
  // descriptors are wanted): the MFCC class that also leaves the whole magnitude spectrum in a.mag_out -- every row
  // untangled in pairs, bins 0..511 from the direct halves, 513..1023 from the mirrored ones, flushed like the
  // reference's -- plus the two spectrum bands above bin 737.  It keeps no sums across rows: the spectral statistics
  // are bands_kernel's (seven sums more in a kernel that has the rows in its registers anyway), the amplitude of the
  // hop hop_kernel's.  The statistics of the full classes cost this kernel 250-300 bytes of scratch per lane and more
  // than half of its time.
  // FEAT 6 (round 6): the magnitude class for masks whose later kernels read bins 0..767 only (bands_kernel; no whitening
  // follower, no magnitude output: !kFramesWholeSpectrum) -- the mirrored blocks of rows 0..7 (bins 769..1023) are not
  // stored and their square roots not taken (spectrum bands 26 / 27 are sums of |X|^2): 6 KiB of stores per frame instead
  // of 8 in a kernel that runs at the memory system's rate.
  constexpr bool MAGS = (FEAT == 4 || FEAT == 6);
  constexpr bool MAGS_UPPER = (FEAT == 4);
  // The overlap half of a frame (rows 0..15 = the hop of the frame before) either stays in 32 registers from frame to
  // frame (MFCC class: it has them) or comes back from the cache by a second LDS-DMA into the upper half of the wave's
  // plane, asked for at the bottom of the loop with the window pairs: the classes that keep sums or stored rows next to
  // the FFT registers spilled those 32 registers (128 B of scratch per lane and frame pair, written and read back: 4 KiB
  // per frame each way -- which the counters showed as HBM traffic, the scratch lines do not survive in L2 next to the
  // streaming stores).
  constexpr int kMelEarly = 4;   // magnitude class: group of the untangle behind which the mel weights are asked for
  // (measured: magnitude class 2.35 -> 1.99 ms on the C4 share, full classes 23.1 -> 20.7 ms on 5.12 M frames; the
  // statistics class, which spilled 44 bytes, 16.1 -> 17.7 ms and the MFCC class 493 -> 422 M frames/s: they keep the
  // registers)
  constexpr bool LO_LDS = (FEAT >= 2 && FEAT <= 4) || FEAT == 6;
  // Cache policy of the classes that fetch every hop twice (round 5, A/B on the C4 share, profiles/r05/ab_cache_policy.txt):
  // the first read without the non-temporal hint -- the line is asked for again one frame later -- and the magnitude
  // stores with it (8 KiB per frame streaming through L2 would push those lines out): 2.14 -> 2.02 ms.
  constexpr bool kHopNt = !LO_LDS;   // first read of a hop
  constexpr bool kLoNt = true;       // its second read, as the next frame's overlap half
  // FEAT 5: the statistics class for masks without skewness / kurtosis -- the descriptor set BASELINE.json's north star
  // names (MFCC + rms, centroid, spread, rolloff, flatness) -- keeps no third and fourth moments: four registers and two
  // operations per magnitude less.
  constexpr bool STATS = (FEAT >= 1 && FEAT <= 3) || FEAT == 5;
  constexpr bool MOMENTS34 = (FEAT != 5);
  constexpr bool PAIRS = STATS;
  // FEAT 2, 3 ("full" classes, for masks with flux / spectrum bands / sub-band descriptors / amplitude): the statistics
  // class that also leaves the magnitudes in a.mag_out for bands_kernel and the whitening kernels, the amplitude peak /
  // rms of the hop (SA:1760-1783) and the two spectrum bands above the analysis range (bins 738..904, 905..1023: sums
  // of |X|^2 straight from the mirrored halves of rows 0..8, no square root).  FEAT 2 stores what the statistics class
  // computes anyway: bins 0..768, all bands_kernel reads.  FEAT 3 also produces and stores the mirrored blocks of rows
  // 0..7 (bins 769..1023): the whole spectrum, for the whitening follower / fail-safe f0 and the magnitude output.
  constexpr bool STORE = (FEAT == 2 || FEAT == 3);
  constexpr bool UPPER = (FEAT == 3);
  // rows of 32 bins that are untangled directly: bins 0..383 for the mel filters, 0..511 (+ their mirrored blocks) for
  // the spectral statistics
  constexpr int MR = (FEAT >= 1) ? 16 : kMel32Rows;
  using Map = Lds32<kMel32Rows>;   // untangle factors of rows 0..11; rows 12..15 are rows 0..3 times w2048^384
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: SGPR arithmetic
  const int h = lane >> 5, q = lane & 31;

  // ---- shared tables ----
  copy_lds_table(lds_raw + Map::tw, a.tw32, 32 * 32 * 16, threadIdx.x, kWaves32 * 64);
  copy_lds_table(lds_raw + Map::post, a.post32, kMel32Rows * 32 * 16, threadIdx.x, kWaves32 * 64);
  copy_lds_table(lds_raw + Map::dct, a.dct, 14 * 16 * 8, threadIdx.x, kWaves32 * 64);
  copy_lds_table(lds_raw + Map::logc, kLogConst, 32 * 8, threadIdx.x, kWaves32 * 64);
#if AFX_STAMPS
  unsigned stamp_acc[16] = {};
  unsigned long long life_core0, life_real0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(life_core0), "=s"(life_real0));
  const bool stamping = (wave == 0) && a.stamps != nullptr;
  unsigned long long stamp_prev = 0;
#endif
  __syncthreads();

  // window and mel tables: buffer loads (one descriptor in SGPRs, lane offset in one VGPR, the row as an
  // immediate / scalar offset) instead of one 64-bit address pair per row
  const __amdgpu_buffer_rsrc_t win_rs = table_rsrc(a.win32, 32 * 32 * 16);                  // [32 n1 + q] double2
  const __amdgpu_buffer_rsrc_t mel_rs = table_rsrc(a.melw32, kMel32Pairs * 32 * 8);         // [32 pair + q] double
  const int q16 = 16 * q, q8 = 8 * q;
  const double2* const tw = reinterpret_cast<const double2*>(lds_raw + Map::tw) + q;        // + 32 n2
  const double2* const post = reinterpret_cast<const double2*>(lds_raw + Map::post) + q;    // + 32 r
  const double* const dct = reinterpret_cast<const double*>(lds_raw + Map::dct);
  const double* const logc = reinterpret_cast<const double*>(lds_raw + Map::logc);
  const int stride2 = 2 * a.lay.stride;
  double* const plane = reinterpret_cast<double*>(lds_raw + Map::xchg + wave * kPlane32Bytes) + kHalfSlots * h;
  double* const pw = plane + q;                      // + 33 k1: this lane is n2 = q when writing
  // + n2: this lane is k1 = q when reading (volatile: one ds_read_b64 per value, never ds_read2_b64 at half
  // the rate; the explicit LDS address space keeps the volatile accesses DS instructions)
  using lds_vdouble = volatile __attribute__((address_space(3))) double;
  lds_vdouble* const pr = (lds_vdouble*)(plane + 33 * q);
  const int partner = (lane & 32) + ((32 - q) & 31);
  // the new hop of the next frame (rows 16..31 of both halves, 8 KiB per wave) is moved global -> LDS by eight
  // global_load_lds_dwordx4 (1 KiB each: lane L's 16 bytes land at 16 L) into the exchange plane, which is idle
  // between two exchanges: image [row][half][q] float2.  Lane L moves complex samples 2 (L & 15), +1 of row
  // 16 + 2 j + (L >> 5) of half (L >> 4) & 1.
  const int dh = (lane >> 4) & 1, drow = lane >> 5, dq = 2 * (lane & 15);
  typedef __attribute__((address_space(3))) void lvoid;
  unsigned char* const plane_bytes = lds_raw + Map::xchg + wave * kPlane32Bytes;
  // the plane's LDS byte address as a wave-uniform scalar (M0 of the DMA)
  const unsigned plane_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lvoid*)plane_bytes);
  // scalar constants of the magnitude (laundered: the compiler must keep them in SGPR pairs, not fold them into VGPRs)
  double k_tiny = 2.2250738585072014e-308, k_three = 3.0;
  unsigned long long k_lane0 = 0x0000000100000001ull;   // EXEC of keep_lane0: lanes 0 and 32
  asm volatile("" : "+s"(k_tiny), "+s"(k_three), "+s"(k_lane0));
  lds_vdouble* const hop = (lds_vdouble*)(plane_bytes + 256 * h + 8 * q);   // + 64 r (8-byte units: 512 bytes a row)

  const float* const pcm = reinterpret_cast<const float*>(a.pcm);

  // Work queue.  A static split is badly unbalanced here: of the two waves that share a SIMD the older one wins
  // the issue arbitration and finishes its share ~35 % earlier (measured), and CUs differ by up to 30 %.  Every
  // wave starts with the pair of chunks of its own index (no atomic on the way in) and then draws further pairs
  // from one device counter.  A draw follows every processed pair, so a launch advances the counter by exactly
  // the number of pairs: the host passes the counter's value at launch (queue_base) instead of resetting it.
  const int n_items = (a.n_chunks + 1) >> 1;
  const int n_static = min((int)(gridDim.x * kWaves32), n_items);
  for (int slot = blockIdx.x * kWaves32 + wave; slot < n_items;) {
    const int ci = 2 * slot + h;
    const bool have = ci < a.n_chunks;
    const Chunk ch = a.chunks[have ? ci : 2 * slot];
    const int nfr = have ? ch.nframes : 0;
    const double sc = ch.scale;   // SCALED: FinalScaling of this half's buffer (the arena holds LoadSample's float signal)
    const int total = max(__builtin_amdgcn_readlane(nfr, 0), __builtin_amdgcn_readlane(nfr, 32));
    const float2* src = reinterpret_cast<const float2*>(pcm + ch.sample_off) + q;
    // this lane's part in the LDS-DMA: the chunk of half dh
    const int cid = 2 * slot + dh;
    const bool have_d = cid < a.n_chunks;
    const Chunk chd = a.chunks[have_d ? cid : 2 * slot];
    const int last_d = max((have_d ? (int)chd.nframes : 0) - 1, 0);
    const float* const dsrc = pcm + chd.sample_off + 64 * (16 + drow) + 2 * dq;   // row 16 + drow, sample 2 dq

    // (pairs of floats as the 8-byte words they are loaded as: an array of float2 was left in scratch memory by the
    // compiler in the larger classes -- 128 bytes per lane stored and reloaded every frame pair)
    double lo[16];
    if constexpr (!LO_LDS) {
      const double* const src8 = reinterpret_cast<const double*>(src);
#pragma unroll
      for (int r = 0; r < 16; ++r) lo[r] = src8[32 * r];
    } else {
      dma_hop<kLoNt>(dsrc - kHop, plane_lds + 8192);   // rows 0..15 of frame 0
    }
    // frame 0's new hop (every DS operation of the previous chunk has been waited for)
    dma_hop<kHopNt>(dsrc, plane_lds);

    double mel_acc = 0.0;  // mel sums of up to two finished frames per half: lane 2 f + slot
    // this lane's output element: coefficient (q >> 1) of the frame in slot q & 1
    double* recp = a.rec + ((int64_t)ch.frame0 + (q & 1)) * a.lay.stride + a.lay.mfcc + ((q >> 1) & 15);
    int left = nfr - (q & 1);

    // The loop body is software-pipelined by hand (two waves per SIMD leave little for the hardware to hide):
    // every table / LDS / cross-lane read is issued one step before the arithmetic that consumes it, and
    // sched_barriers keep the compiler from folding the steps back together.
    double2 w[32];   // window pairs of the NEXT frame: loaded at the bottom of the loop, when the FFT registers are dead
#pragma unroll
    for (int r = 0; r < 32; ++r) w[r] = table_load2(win_rs, q16, 512 * r);

#if AFX_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev));
#endif
    for (int fi = 0; fi < total; ++fi) {
      // ---- window (table carries the 1/2048 of kDivFwdByN and the 1/2 of the untangle) ----
      cx<double> v[32];
      {
        double nx[16];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the hop's LDS-DMA has landed (and the window pairs)
        AFX_STAMP(0);   // wait for DMA + window
#pragma unroll
        for (int r = 0; r < 16; ++r) nx[r] = hop[64 * r];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LO_LDS) {
          const double* const lo_rows = reinterpret_cast<const double*>(plane_bytes + 8192 + 256 * h) + q;
#pragma unroll
          for (int r = 0; r < 16; ++r) lo[r] = lo_rows[64 * r];
        }
        // window products fused into the first radix-4 stage of P1 (rows j, j + 8 from the overlap half, rows
        // j + 16, j + 24 from the new hop): a w_a +- c w_c as one product and two fused multiply-adds
        double hop_sq = 0.0, hop_max = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          auto first = [](double pair) { return __int_as_float(__double2loint(pair)); };
          auto second = [](double pair) { return __int_as_float(__double2hiint(pair)); };
          const double ar = pcm_double<SCALED>(first(lo[j]), sc), ai = pcm_double<SCALED>(second(lo[j]), sc);
          const double br = pcm_double<SCALED>(first(lo[j + 8]), sc), bi = pcm_double<SCALED>(second(lo[j + 8]), sc);
          const double cr = pcm_double<SCALED>(first(nx[j]), sc), ci = pcm_double<SCALED>(second(nx[j]), sc);
          const double dr = pcm_double<SCALED>(first(nx[j + 8]), sc), di = pcm_double<SCALED>(second(nx[j + 8]), sc);
          if constexpr (MAGS) {   // the hop (rows 0..15 of the frame) is in registers here: its amplitude (SA:1760-1783)
            hop_sq = fma(ar, ar, fma(ai, ai, fma(br, br, fma(bi, bi, hop_sq))));
            hop_max = fmax(fmax(hop_max, fmax(fabs(ar), fabs(ai))), fmax(fabs(br), fabs(bi)));
          }
          const double par = ar * w[j].x, pai = ai * w[j].y, pbr = br * w[j + 8].x, pbi = bi * w[j + 8].y;
          const double t0r = fma(cr, w[j + 16].x, par), t0i = fma(ci, w[j + 16].y, pai);
          const double t1r = fma(-cr, w[j + 16].x, par), t1i = fma(-ci, w[j + 16].y, pai);
          const double t2r = fma(dr, w[j + 24].x, pbr), t2i = fma(di, w[j + 24].y, pbi);
          const double t3r = fma(-dr, w[j + 24].x, pbr), t3i = fma(-di, w[j + 24].y, pbi);
          v[j] = {t0r + t2r, t0i + t2i};
          v[j + 16] = {t0r - t2r, t0i - t2i};
          v[j + 8] = {t1r + t3i, t1i - t3r};
          v[j + 24] = {t1r - t3i, t1i + t3r};
        }
        if constexpr (!LO_LDS) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lo[r] = nx[r];
        }
        if constexpr (MAGS) {
          // amplitude_peak / amplitude_rms of the hop, reduced over the half's 32 lanes (magnitude class: no hop_kernel
          // launch and no second pass over the PCM for masks without f0)
          if (a.mask & ((1u << 11) | (1u << 12))) {   // AFX_D_AMPLITUDE_PEAK | AFX_D_AMPLITUDE_RMS
            hop_sq += dpp_or_zero<kDppXor1>(hop_sq);      hop_max = fmax(hop_max, dpp_mov<kDppXor1>(hop_max));
            hop_sq += dpp_or_zero<kDppXor2>(hop_sq);      hop_max = fmax(hop_max, dpp_mov<kDppXor2>(hop_max));
            hop_sq += dpp_or_zero<kDppHalfMirror>(hop_sq); hop_max = fmax(hop_max, dpp_mov<kDppHalfMirror>(hop_max));
            hop_sq += dpp_or_zero<kDppMirror>(hop_sq);    hop_max = fmax(hop_max, dpp_mov<kDppMirror>(hop_max));
            // the two rows of a half: lane 0 / 16 of half 0, 32 / 48 of half 1
            const double sq_h = h ? read_lane<32>(hop_sq) + read_lane<48>(hop_sq) : read_lane<0>(hop_sq) + read_lane<16>(hop_sq);
            const double mx_h = h ? fmax(read_lane<32>(hop_max), read_lane<48>(hop_max)) : fmax(read_lane<0>(hop_max), read_lane<16>(hop_max));
            if (q == 0 && fi < nfr) {
              double* const rec = a.rec + ((int64_t)ch.frame0 + fi) * a.lay.stride;
              if ((a.mask & (1u << 11)) && a.lay.amp_peak >= 0) rec[a.lay.amp_peak] = mx_h;
              if ((a.mask & (1u << 12)) && a.lay.amp_rms >= 0) rec[a.lay.amp_rms] = nan_to_zero(sqrt(sq_h / (double)kHop));
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      AFX_STAMP(1);   // hop reads + conversion + window + first radix-4 stage
      // ---- P1 (rest) ----
      f32x32::dft32_rest(v);
      __builtin_amdgcn_sched_barrier(0);
      AFX_STAMP(2);   // P1

      // ---- E: real parts, then imaginary parts through the same plane (a wave's DS operations execute in
      // order, so the second round of writes needs no wait for the first round of reads); the first two groups
      // of twiddles queue up behind them ----
      // twiddles by first-stage butterfly: group g = butterflies j = 2 g, 2 g + 1 on rows j, j + 8, j + 16, j + 24
      double2 t[32];
      auto load_tw = [&](int g) {
#pragma unroll
        for (int j = 2 * g; j < 2 * g + 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (j + 8 * i != 0) t[j + 8 * i] = tw[32 * (j + 8 * i)];
      };
      auto bfly_tw = [&](int g) {
#pragma unroll
        for (int j = 2 * g; j < 2 * g + 2; ++j) {
          const cx<double> w1{t[j + 8].x, t[j + 8].y}, w2{t[j + 16].x, t[j + 16].y}, w3{t[j + 24].x, t[j + 24].y};
          if (j == 0) f32x32::radix4_tw3(v[0], v[8], v[16], v[24], w1, w2, w3);
          else f32x32::radix4_tw4(v[j], v[j + 8], v[j + 16], v[j + 24], cx<double>{t[j].x, t[j].y}, w1, w2, w3);
        }
      };
      {
        double re[32], im[32];
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) pw[33 * k1] = v[k1].re;
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) re[n2] = pr[n2];
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) pw[33 * k1] = v[k1].im;
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) im[n2] = pr[n2];
        load_tw(0);
        load_tw(1);
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) v[n2] = {re[n2], im[n2]};
      }
      // ---- the next frame's new hop: LDS-DMA into the plane once the exchange reads have returned (the DMA is a
      // vector-memory operation: nothing else orders it behind this wave's DS reads).  The last frame of a chunk
      // re-reads its own hop (no branch: the loop body stays one basic block).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      AFX_STAMP(3);   // exchange
      dma_hop<kHopNt>(dsrc + (size_t)min(fi + 1, last_d) * kHop, plane_lds);
      __builtin_amdgcn_sched_barrier(0);

      // ---- T + P2: v[k2] = Z[q + 32 k2]; the factors w1024^(n2 k1) are fused into the first radix-4 stage of
      // the 32-point DFT, two butterflies at a time, the twiddles of the butterflies after next in flight ----
      bfly_tw(0);
      load_tw(2);
      __builtin_amdgcn_sched_barrier(0);
      bfly_tw(1);
      load_tw(3);
      __builtin_amdgcn_sched_barrier(0);
      bfly_tw(2);
      bfly_tw(3);
      __builtin_amdgcn_sched_barrier(0);
      AFX_STAMP(4);   // DMA issue + twiddled first stage
      f32x32::dft32_rest(v);
      __builtin_amdgcn_sched_barrier(0);
      AFX_STAMP(5);   // P2

      // ---- U: untangle + magnitude: mag[r] = |X[q + 32 r]|, four rows at a time, the next four rows' partner
      // values (cross-lane) and post-twiddles (LDS) in flight ----
      double mag[MR];
      cx<double> pp[MR];
      double2 pw2[MR];
      double mwt[kMel32Pairs];
      // statistics class: per-lane partial sums over this lane's bins of the analysis range (bins 1..738, j = bin - 1):
      // sum m, m^2, j m, j^2 m, m^3, m^4 and the product of (m + 1e-20) in two halves (24 factors would underflow)
      double s1 = 0.0, s2 = 0.0, sj = 0.0, sjj = 0.0, s3 = 0.0, s4 = 0.0, prod_a = 1.0, prod_b = 1.0;
      double jq = 0.0, pa = 0.0;
      double jmir = 0.0;   // pairs: j of this lane's bin in the next mirrored block: 1023 - q - 32 r, r = 8, 9, ..
      double mir[4] = {0.0, 0.0, 0.0, 0.0}, cmid = 0.0;   // pairs: M_12..M_15 (index r - 12) and bin 512 (lane 0)
      // statistics class: magnitudes of rows 0..11 wait here for the rolloff walk (upper part of the wave's exchange
      // plane: 384 doubles per half behind the 8 KiB the hop DMA writes)
      double* const park = reinterpret_cast<double*>(plane_bytes + 8192 + 4352 * h);
      // full classes: this half's row of a.mag_out; a frame past the end of the half's chunk writes the spare row behind
      // the last frame instead (no branch around the stores).  Stored magnitudes are flushed like the reference's
      // (TAudioMath::Magnitude runs with DAZ + FZ): exactly 0 at or below kFlushLevel -- the value the sums take anyway.
      // The address is rebuilt at every group of stores: two pointers kept for the whole frame are four registers
      // this kernel does not have.
      auto mag_row = [&]() -> double* {
        int fl = fi, ql = lane;
        asm volatile("" : "+v"(ql), "+s"(fl));
        const int64_t mrow = (fl < nfr) ? (int64_t)ch.frame0 + fl : a.mag_spare_row;
        return a.mag_out + mrow * 1024 + (ql & 31);          // + 32 r: bin 32 r + q;  + 1024 - 2 q - 32 r: bin 1024 - 32 r - q
      };
      double band26 = 0.0, band27 = 0.0;   // full classes: sums of |X|^2 over bins 738..904 and 905..1023 (spectrum bands 26, 27)
      // magnitude class: the two pointers of this half's row (+ 32 r: bin 32 r + q; - 32 r: bin 1024 - 32 r - q), rebuilt
      // before every group of rows, and the store that flushes like TAudioMath::Magnitude under DAZ + FZ
      double* rowd = nullptr;
      double* rowm = nullptr;
      auto mag_rows = [&]() {
        rowd = mag_row();
        int ql = lane;
        asm volatile("" : "+v"(ql));
        rowm = rowd + 1024 - 2 * (ql & 31);
      };
      auto store_mag = [&](double* p, double value) { __builtin_nontemporal_store(value > logc[kCSqrtMin] ? value : 0.0, p); };
      if (STATS) {
        int ql = lane;
        asm volatile("" : "+v"(ql));
        jq = (double)((ql & 31) - 1);
        if (PAIRS) jmir = (double)(767 - (ql & 31));   // M_8: bin 1024 - 256 - q
      }
      // statistics class: one value's contribution to the sums over bins 1..738 (j = bin - 1): the direct rows 0..15
      // (row 0 without bin 0; rows 0..11 after the mel stage, when the registers of the FFT are free, rows 12..15 as
      // they are produced), bin 512, and the mirrored blocks M_15..M_8 (of M_8 = bins 737..768 only two bins).
      // Magnitudes below sqrt(DBL_MIN) are 0 in the reference (TAudioMath::Magnitude runs with DAZ + FZ): a silent
      // frame has centroid 0, not (n - 1) / 2.
      // accumulate_at: a value that is not one of the direct rows (mirrored block, bin 512)
      // store_at: full classes, where the flushed value goes (nullptr: nowhere); every in-range caller stores what it sums
      auto accumulate_at = [&](double value, double j, bool in_range, double* store_at, bool all_in_range) {
        const bool audible = value > logc[kCSqrtMin];
        const bool ok = in_range && audible;
        const double m = ok ? value : 0.0;
        if (STORE && store_at) *store_at = all_in_range ? m : (audible ? value : 0.0);
        const double m2 = m * m;
        const double jm = j * m;
        s1 += m;
        s2 += m2;
        sj += jm;
        sjj = fma(j, jm, sjj);
        if (MOMENTS34) {
          s3 = fma(m2, m, s3);
          s4 = fma(m2, m2, s4);
        }
        prod_b *= ok ? (value + logc[kCEps]) : 1.0;
      };
      auto accumulate = [&](int r, double* store_at) {
        const bool audible = mag[r] > logc[kCSqrtMin];
        const bool ok = ((r == 0) ? (q != 0) : true) && audible;
        const double m = ok ? mag[r] : 0.0;
        if (STORE && store_at) *store_at = (r == 0) ? (audible ? mag[r] : 0.0) : m;   // bin 0 is outside the sums, not outside the spectrum
        const double m2 = m * m;
        const double jm = jq * m;
        s1 += m;
        s2 += m2;
        sj += jm;
        sjj = fma(jq, jm, sjj);
        if (MOMENTS34) {
          s3 = fma(m2, m, s3);
          s4 = fma(m2, m2, s4);
        }
        const double f = ok ? (mag[r] + logc[kCEps]) : 1.0;
        if (r < 12) prod_a *= f; else prod_b *= f;
        jq += logc[kC32];
      };
      auto fetch = [&](int r) {
        pp[r] = {__shfl(v[31 - r].re, partner), __shfl(v[31 - r].im, partner)};
        pw2[r] = post[32 * (r % kMel32Rows)];
      };
      auto untangle = [&](int r) {
        cx<double> p = pp[r];
        keep_lane0(p.re, p.im, v[(32 - r) & 31].re, v[(32 - r) & 31].im, k_lane0);   // q == 0: bin 32 r pairs with this lane's Z[1024 - 32 r]
        const cx<double> z = v[r];
        double2 wq = pw2[r];
        if (r >= kMel32Rows) {   // w2048^(q + 32 r) = w2048^(q + 32 (r - 12)) w2048^384
          const double cr = logc[kCRotRe], ci = logc[kCRotIm];
          wq = double2{fma(wq.x, cr, -wq.y * ci), fma(wq.x, ci, wq.y * cr)};
        }
        const double er = z.re + p.re, ei = z.im - p.im;   // E = Z + conj(P)
        const double orr = z.im + p.im, oi = p.re - z.re;  // O = -i (Z - conj(P))
        if (MAGS) {
          const double tr = fma(wq.x, orr, -wq.y * oi), ti = fma(wq.x, oi, wq.y * orr);   // w O
          const double xr = er + tr, xi = ei + ti, yr = er - tr, yi = ei - ti;
          const double md = mag_sqrt_mel(xr, xi, k_tiny, k_three);                        // |X[32 r + q]|
          const double mv = mag_sqrt_mel(yr, yi, k_tiny, k_three);                        // |X[1024 - 32 r - q]|
          if (r < kMel32Rows) park[32 * r + q] = md;    // rows 0..11 wait in LDS for the mel stage (no register to spare here)
          store_mag(rowd + 32 * r, md);
          if (MAGS_UPPER || r >= 8) {                                                      // bins 1024 - 32 r - q: above 768 for r < 8
            if (!(r == 0 && q == 0)) store_mag(rowm - 32 * r, mv);                        // (row 0, lane 0 would be bin 1024)
          }
          // spectrum bands 26 = bins 738..904 and 27 = bins 905..1023 lie in the mirrored halves of rows 0..8
          if (r <= 8) {
            const double sq = fma(yi, yi, yr * yr);
            if (r < 3) band27 += (r == 0 && q == 0) ? 0.0 : sq;
            else if (r == 3) { band27 += (q <= 23) ? sq : 0.0; band26 += (q <= 23) ? 0.0 : sq; }
            else if (r < 8) band26 += sq;
            else band26 += (q <= 30) ? sq : 0.0;                                          // M_8 = bins 737..768
          }
        } else if (PAIRS && (STORE || r >= 8)) {
          const double tr = fma(wq.x, orr, -wq.y * oi), ti = fma(wq.x, oi, wq.y * orr);   // w O
          const double xr = er + tr, xi = ei + ti, yr = er - tr, yi = ei - ti;
          mag[r] = mag_sqrt_mel(xr, xi, k_tiny, k_three);
          if (r < 8) {
            // the mirrored block lies above the analysis range (bins 769..1023): |X|^2 into the two spectrum bands there
            // (M_3 = bins 897..928 holds the edge at 905); row 0, lane 0 would be bin 1024, which does not exist
            const double sq = fma(yi, yi, yr * yr);
            if (r < 3) band27 += (r == 0 && q == 0) ? 0.0 : sq;
            else if (r == 3) { band27 += (q <= 23) ? sq : 0.0; band26 += (q <= 23) ? 0.0 : sq; }
            else band26 += sq;
            if (UPPER) {
              const double mv = mag_sqrt_mel(yr, yi, k_tiny, k_three);
              double* const dst = mag_row() + (1024 - 32 * r) - 2 * q;
              if (!(r == 0 && q == 0)) *dst = mv > logc[kCSqrtMin] ? mv : 0.0;
            }
          } else {
            const double mv = mag_sqrt_mel(yr, yi, k_tiny, k_three);                      // |X[1024 - k]|
            if (r < kMel32Rows) park[32 * (r + 4) + q] = mv;    // M_8..M_11 wait in LDS (slots 12..15) for the registers of the FFT
            else {
              mir[r - kMel32Rows] = mv;
              accumulate_at(mv, jmir, true, STORE ? mag_row() + (1024 - 32 * r) - 2 * q : nullptr, true);
              jmir -= logc[kC32];
            }
          }
        } else {
          const double xr = fma(wq.x, orr, fma(-wq.y, oi, er));
          const double xi = fma(wq.x, oi, fma(wq.y, orr, ei));
          mag[r] = mag_sqrt_mel(xr, xi, k_tiny, k_three);
        }
        if (STATS && r >= kMel32Rows) accumulate(r, STORE ? mag_row() + 32 * r : nullptr);
      };
      if constexpr (MAGS) {
        // magnitude class: sixteen rows in pairs of two, one pair of fetches ahead (the paired form keeps more live per
        // row than the other classes' and every value leaves at once: short groups keep the register file below its limit)
        fetch(0); fetch(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (g < 7) { fetch(2 * g + 2); fetch(2 * g + 3); }
          mag_rows();
          untangle(2 * g);
          untangle(2 * g + 1);
          // The packed mel rows, asked for in the MIDDLE of the untangle: on gfx950 stores count in vmcnt like loads and
          // retire in order, so weights asked for behind the frame's last store (where the last FFT registers die) made the
          // mel stage wait for every store of the frame.  Half-way through, the rows already untangled have freed 16
          // registers per group; the loads then wait only for the stores issued before them, which are long done when
          // the mel stage starts, and the compiler's wait count lets the later stores stay in flight.
          if (g == kMelEarly) {
#pragma unroll
            for (int i = 0; i < kMel32Pairs; ++i) mwt[i] = table_load1(mel_rs, q8, 256 * i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // bin 512 = lane 0 of row 16 is its own partner: E = 2 Re Z, w O = -2i Im Z (the window carries the 1/2)
        const double er = v[16].re + v[16].re, orr = v[16].im + v[16].im;
        const double cm = mag_sqrt_mel(er, orr, k_tiny, k_three);
        if (q == 0) store_mag(rowd + 512, cm);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < kMel32Rows; ++r) mag[r] = park[32 * r + q];
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (STATS) {
        // statistics and full classes: rows in groups of two, one group of fetches ahead.  They keep eighteen doubles of sums
        // next to the FFT registers; with four rows in flight (as below) the allocator spilled 88 bytes per lane, with two 36,
        // and the shorter groups cost less than the spills did (285 -> 291 M frames/s on the star set).
        fetch(0); fetch(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          if (g < 5) { fetch(2 * g + 2); fetch(2 * g + 3); }
          untangle(2 * g);
          untangle(2 * g + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) fetch(r);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 4; r < 8; ++r) fetch(r);
#pragma unroll
      for (int r = 0; r < 4; ++r) untangle(r);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 8; r < 12; ++r) fetch(r);
#pragma unroll
      for (int r = 4; r < 8; ++r) untangle(r);
      __builtin_amdgcn_sched_barrier(0);
      // the packed mel rows come from global memory (L1/L2-resident): issued here, when two thirds of the FFT
      // registers are dead
      if (FEAT == 0) {
#pragma unroll
        for (int i = 0; i < kMel32Pairs; ++i) mwt[i] = table_load1(mel_rs, q8, 256 * i);
      }
#pragma unroll
      for (int r = 8; r < 12; ++r) untangle(r);
      __builtin_amdgcn_sched_barrier(0);
      }
      if (STATS) {   // statistics class: behind the last rows (62 registers in flight that it cannot spare earlier)
#pragma unroll
        for (int i = 0; i < kMel32Pairs; ++i) mwt[i] = table_load1(mel_rs, q8, 256 * i);
      }
      AFX_STAMP(6);   // untangle

      // ---- MFCC: sparse mel rows now; log + DCT once per two iterations (vector.c:350-391) ----
      {
        double e[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) e[f] = 0.0;
#pragma unroll
        for (int r = 0; r < kMel32Rows; ++r)
#pragma unroll
          for (int f = 0; f < kNumCep; ++f)
            if (mel32_touches(f, r)) e[f] += mag[r] * mwt[mel32_pair_index(r, f)];
        __builtin_amdgcn_sched_barrier(0);
        AFX_STAMP(7);   // mel rows (waits for the table loads)
        if (STATS) {
          // statistics class: the mel sums are reduced first (their registers are needed), then rows 12..15 and their mirrors
          const double tot1 = half_sum16(e, lane);
          if ((lane & 1) == (fi & 1)) mel_acc = tot1;
          {
            double* const row0 = STORE ? mag_row() : nullptr;
#pragma unroll
            for (int r = 0; r < kMel32Rows; ++r) accumulate(r, STORE ? row0 + 32 * r : nullptr);
          }
          // rows 0..11 are needed once more (rolloff): their row sums now, the values parked in the part of the
          // exchange plane that the hop DMA does not use
          {
            double ra[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[r] = (r < 12) ? ((r == 0 && q == 0) ? 0.0 : mag[r]) : 0.0;
            pa = half_sum16(ra, lane);               // lane L: row (L & 31) >> 1
#pragma unroll
            for (int r = 0; r < 12; ++r) park[32 * r + q] = mag[r];

          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 12; r < 16; ++r) fetch(r);
          if constexpr (PAIRS) {   // the mirrored blocks of rows 8..11, parked by the untangle stage: bins 737..768 (two of them in range) .. 641..672
            double* const mrow0 = STORE ? mag_row() + 1024 - 2 * q : nullptr;   // bin 1024 - 32 r - q at - 32 r
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const double mv = park[32 * (12 + i) + q];
              // M_8 = bins 737..768: two of them inside the analysis range, the others (q <= 30) in spectrum band 26
              if (STORE && i == 0) band26 += (q <= 30) ? 0.25 * (mv * mv) : 0.0;   // (the row sums are of the halved spectrum, x 4 below)
              accumulate_at(mv, jmir, (i == 0) ? (q >= 30) : true, STORE ? mrow0 - 32 * (8 + i) : nullptr, i != 0);
              jmir -= logc[kC32];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 12; r < 16; ++r) untangle(r);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (PAIRS) {
            // bin 512 = lane 0 of row 16 is its own partner: E = 2 Re Z, w O = -2i Im Z (the window carries the 1/2)
            const double er = v[16].re + v[16].re, orr = v[16].im + v[16].im;
            cmid = (q == 0) ? mag_sqrt_mel(er, orr, k_tiny, k_three) : 0.0;
            accumulate_at(cmid, logc[kC511], q == 0, (STORE && q == 0) ? mag_row() + 512 : nullptr, true);
          }
        }
        // the next frame's window pairs, under the reduction and the log / DCT (statistics class: behind its sums,
        // which need the registers)
        if (FEAT == 0 || MAGS) {
#pragma unroll
          for (int r = 0; r < 32; ++r) w[r] = table_load2(win_rs, q16, 512 * r);
          // (the parked rows have been read: the upper half of the plane is free for the next frame's overlap rows)
          if constexpr (LO_LDS) dma_hop<kLoNt>(dsrc + (size_t)min(fi + 1, last_d) * kHop - kHop, plane_lds + 8192);
        }
        if (FEAT == 0 || MAGS) {
          if (MAGS) {   // the two free slots of the reduction: spectrum bands 26, 27 (x 4: the window carries an extra 1/2)
            e[14] = 4.0 * band26;
            e[15] = 4.0 * band27;
          }
          const double tot = half_sum16(e, lane);  // lane L: filter (L & 31) >> 1 of this half's frame
          if ((lane & 1) == (fi & 1)) mel_acc = tot;
          if (MAGS) {
            int ql = lane;
            asm volatile("" : "+v"(ql));
            const int hq = ql & 31;
            double* const rec = a.rec + ((int64_t)ch.frame0 + fi) * a.lay.stride;
            if (fi < nfr && a.lay.bands >= 0 && (hq == 28 || hq == 30)) rec[a.lay.bands + 26 + ((hq - 28) >> 1)] = tot;
          }
        }
        AFX_STAMP(8);   // window issue + reduction
        if (STATS) {
          // ---- spectral statistics (SA:1808-1933): the sums of this half's frame, reduced over its 32 lanes; the
          //      closed forms (sqrt, divisions, exp / log of the flatness) are left to stats32_finish_kernel ----
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const int hq = ln & 31;
          double st[16];
          st[0] = s1; st[1] = s2; st[2] = sj; st[3] = sjj; st[4] = s3; st[5] = s4;
          st[6] = log_lds(prod_a, logc) + log_lds(prod_b, logc);
          st[7] = STORE ? 4.0 * band26 : 0.0;   // the window carries an extra 1/2 (mag_sqrt_mel): |X|^2 = 4 |X / 2|^2
          st[8] = STORE ? 4.0 * band27 : 0.0;
#pragma unroll
          for (int i = 9; i < 16; ++i) st[i] = 0.0;
          const double red = half_sum16(st, ln);    // lane L: st[(L & 31) >> 1]
          const int64_t row = (int64_t)ch.frame0 + fi;
          const bool live = fi < nfr;
          double* const tmp = a.stat_tmp + row * kStatTmp;
          if (live && (hq & 1) == 0 && (hq >> 1) < 7) tmp[hq >> 1] = red;
          if (STORE) {   // slots 7, 8: spectrum bands 26, 27 straight into the record
            double* const rec = a.rec + row * a.lay.stride;
            if (live && a.lay.bands >= 0 && (hq == 14 || hq == 16)) rec[a.lay.bands + 26 + ((hq >> 1) - 7)] = red;
          }
          // ---- rolloff (scalar.c:472-492): bins whose running sum stays below 85 % of the total ----
          const int base_idx = (ln & 32) << 2;       // byte index of lane 32 h for ds_bpermute
          const double total = __hiloint2double(__builtin_amdgcn_ds_bpermute(base_idx, __double2hiint(red)),
                                                __builtin_amdgcn_ds_bpermute(base_idx, __double2loint(red)));
          const double pivot = total * logc[kCRoll];
          // block sums (rows 0..11: pa, taken before they were parked in LDS); the blocks behind
          // row 11 in bin order are rows 12..15, bin 512, M_15..M_12 (registers), M_11..M_8 (parked): 13 slots
          double rb[16];
          if constexpr (PAIRS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rb[r] = mag[12 + r];
            rb[4] = cmid;
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[5 + i] = mir[3 - i];
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[9 + i] = (i == 3 && hq < 30) ? 0.0 : park[32 * (15 - i) + hq];
            rb[13] = rb[14] = rb[15] = 0.0;
          }
          double pb = half_sum16(rb, ln);            // lane L: slot (L & 31) >> 1
          // inclusive prefix over the row slots of a half (both lanes of a slot hold the same value)
          // (DPP row shifts, then the last lane of a half's first row into its second row: no LDS round trips)
          pa += dpp_or_zero<kDppRowShr2>(pa);  pb += dpp_or_zero<kDppRowShr2>(pb);
          pa += dpp_or_zero<kDppRowShr4>(pa);  pb += dpp_or_zero<kDppRowShr4>(pb);
          pa += dpp_or_zero<kDppRowShr8>(pa);  pb += dpp_or_zero<kDppRowShr8>(pb);
          pa += dpp_or_zero<kDppRowBcast15, 0xA>(pa);  pb += dpp_or_zero<kDppRowBcast15, 0xA>(pb);
          const int last_idx = base_idx + (22 << 2);   // slot 11
          const double tot_a = __hiloint2double(__builtin_amdgcn_ds_bpermute(last_idx, __double2hiint(pa)),
                                                __builtin_amdgcn_ds_bpermute(last_idx, __double2loint(pa)));
          pb += tot_a;
          // rows wholly below the pivot (inclusive prefix < pivot)
          const unsigned long long ma = __ballot((hq & 1) == 0 && (hq >> 1) < 12 && pa < pivot);
          const unsigned long long mb = __ballot((hq & 1) == 0 && (hq >> 1) < 13 && pb < pivot);
          const unsigned halfmask_shift = ln & 32;
          const int rstar = __popc((unsigned)(ma >> halfmask_shift)) + __popc((unsigned)(mb >> halfmask_shift));
          // running sum before row rstar
          const int prev = rstar - 1;
          const int slot = (prev < 12) ? prev : prev - 12;
          const int idx_a = base_idx + ((2 * (slot & 15)) << 2);
          const double base_a = __hiloint2double(__builtin_amdgcn_ds_bpermute(idx_a, __double2hiint(pa)),
                                                 __builtin_amdgcn_ds_bpermute(idx_a, __double2loint(pa)));
          const double base_b = __hiloint2double(__builtin_amdgcn_ds_bpermute(idx_a, __double2hiint(pb)),
                                                 __builtin_amdgcn_ds_bpermute(idx_a, __double2loint(pb)));
          const double before = (rstar == 0) ? 0.0 : ((prev < 12) ? base_a : base_b);
          // the crossing row's magnitudes, summed along the lanes: rows 0..11 from their LDS parking place
          double xs;
          bool okq;
          int bins_before;
          if constexpr (PAIRS) {
            // block rstar in bin order: 0..15 rows, 16 bin 512, 17..20 M_15..M_12 (registers), 21..24 M_11..M_8 (parked);
            // a mirrored block holds its bins in reversed lane order
            const bool parked_m = rstar >= 21;
            const int prow = (rstar < 12) ? rstar : (parked_m ? 36 - rstar : 0);
            xs = park[32 * prow + (parked_m ? 31 - hq : hq)];
#pragma unroll
            for (int r = 12; r < 16; ++r) xs = (rstar == r) ? mag[r] : xs;
            xs = (rstar == 16) ? cmid : xs;
            double xm = mir[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) xm = (rstar == 20 - i) ? mir[i] : xm;
            const int ridx = base_idx + ((31 - hq) << 2);
            const double xrev = __hiloint2double(__builtin_amdgcn_ds_bpermute(ridx, __double2hiint(xm)),
                                                 __builtin_amdgcn_ds_bpermute(ridx, __double2loint(xm)));
            xs = (rstar >= 17 && rstar <= 20) ? xrev : xs;
            okq = (rstar == 0) ? (hq != 0) : ((rstar == 16) ? (hq == 0) : ((rstar == 24) ? (hq <= 1) : (rstar < 24)));
            bins_before = (rstar < 16) ? 32 * rstar - (rstar > 0 ? 1 : 0) : ((rstar == 16) ? 511 : 512 + 32 * (rstar - 17));
          }
          xs = okq ? xs : 0.0;
          const double run = halfwave_scan_incl(xs);
          const unsigned long long mrow = __ballot(okq && (before + run) < pivot);
          const int in_row = __popc((unsigned)(mrow >> halfmask_shift));
          const int below = bins_before + in_row;
          int cnt = (pivot > 0.0) ? below + 1 : 0;
          cnt = cnt > kBinCount ? kBinCount : cnt;
          if (live && hq == 0) tmp[7] = (double)cnt;
          __builtin_amdgcn_sched_barrier(0);

#pragma unroll
          for (int r = 0; r < 32; ++r) w[r] = table_load2(win_rs, q16, 512 * r);
          if constexpr (LO_LDS) dma_hop<kLoNt>(dsrc + (size_t)min(fi + 1, last_d) * kHop - kHop, plane_lds + 8192);
        }
        if (fi & 1) finish_mfcc32(mel_acc, recp, left, stride2, dct, logc, lane);
        AFX_STAMP(9);   // log + DCT + store (every second iteration)
#if AFX_STAMPS
        stamp_acc[15] += 1;
#endif
      }
    }
    if (total & 1) finish_mfcc32(mel_acc, recp, left, stride2, dct, logc, lane);
    unsigned drawn = 0;
    if (lane == 0) drawn = atomicAdd(a.queue, 1u);
    slot = n_static + (int)((unsigned)__builtin_amdgcn_readfirstlane(drawn) - a.queue_base);
  }
#if AFX_STAMPS
  if (stamping && lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&a.stamps[i], (unsigned long long)stamp_acc[i]);
  if (a.stamps != nullptr && lane == 0) {
    // life of this wave: core-clock and 100 MHz real-time stamps at entry and exit, per workgroup
    unsigned long long life_core1, life_real1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(life_core1), "=s"(life_real1));
    unsigned long long* const l = a.stamps + 16 + 4 * ((blockIdx.x & 255) * 8 + wave);
    l[0] = life_core0; l[1] = life_core1; l[2] = life_real0; l[3] = life_real1;
  }
#endif
}

// Closed forms of the spectral statistics from the raw sums the statistics-class kernel left per frame
// (SA:1808-1933; the same formulas as afx_kernels.hip, with the second moments expanded around the centroid):
// tmp[f] = {sum m, sum m^2, sum j m, sum j^2 m, sum m^3, sum m^4, sum log(m + 1e-20), rolloff count, sum x^2 and
// max |x| of the hop (full class)}
__global__ __launch_bounds__(256) void stats32_finish_kernel(const FrameArgs a, int64_t n_frames) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_frames) return;
  const double* t = a.stat_tmp + f * kStatTmp;
  const double s1 = t[0], s2 = t[1], sj = t[2], sjj = t[3], s3 = t[4], s4 = t[5], slog = t[6], cnt = t[7];
  double* const rec = a.rec + f * a.lay.stride;
  const double n = (double)kBinCount, inv_n = 1.0 / (double)kBinCount;
  if (a.mask & (1u << 1)) rec[a.lay.srms] = nan_to_zero(sqrt(s2 * inv_n));
  if (a.mask & 0x3Cu) {
    double cen = 0.0, spr = 0.0;
    if (s1 != 0.0) {
      cen = sj / s1;
      spr = (sjj - cen * sj) / s1;      // sum (j - cen)^2 m = sum j^2 m - cen sum j m
    }
    if (a.lay.centroid >= 0) rec[a.lay.centroid] = cen;
    if (a.lay.spread >= 0) rec[a.lay.spread] = spr;
    if (a.mask & 0x30u) {
      double sk = 0.0, ku = 0.0;
      if (fabs(spr) > (double)1e-12f) {
        const double inv = 1.0 / spr, c2 = cen * cen;
        // sum (m - cen)^3 and sum (m - cen)^4 over the 738 bins
        const double m3 = s3 - 3.0 * cen * s2 + 3.0 * c2 * s1 - n * c2 * cen;
        const double m4 = s4 - 4.0 * cen * s3 + 6.0 * c2 * s2 - 4.0 * c2 * cen * s1 + n * c2 * c2;
        const double i2 = inv * inv;
        sk = m3 * i2 * inv * inv_n;
        ku = m4 * i2 * i2 * inv_n - 3.0;
      }
      if (a.lay.skew >= 0) rec[a.lay.skew] = sk;
      if (a.lay.kurt >= 0) rec[a.lay.kurt] = ku;
    }
  }
  if (a.mask & (1u << 7)) {
    const double gm = fast_exp(slog * inv_n);
    const double am = s1 * inv_n;
    const double fl = (am == 0.0) ? 0.0 : gm / am;
    const double d = lin_to_db(fl) * (-1.0 / 60.0);
    rec[a.lay.flatness] = nan_to_zero(d < 1.0 ? d : 1.0);
  }
  if (a.mask & (1u << 6)) rec[a.lay.rolloff] = cnt * (double)(kSampleRate / (kFft / 2));
}

}  // namespace

int frames32_waves_per_block() { return kWaves32; }
int frames32_stat_tmp_doubles() { return kStatTmp; }

// which (descriptor mask, arithmetic, PCM type) combinations the half-wave kernels serve: MFCC alone; MFCC with any
// of the spectral statistics rms / centroid / spread / skewness / kurtosis / rolloff / flatness (bits 1..7); and the
// full class: those plus the amplitude of the hop (bits 11, 12) and the stored magnitudes (bit 13) that bands_kernel
// turns into flux, the 28 spectrum bands and the sub-band descriptors (bits 8..10)
bool frames_use_halfwave(uint32_t mask, int precision, int pcm_dtype) {
  return (mask & 1u) && !(mask & ~(0x3FFFu | kFramesWholeSpectrum | kFramesStatsLater)) && precision == 0 && (pcm_dtype == kPcmF32 || pcm_dtype == kPcmScaledF32);
}
// 0 = MFCC only, 1 = + spectral statistics, 2 = full with bins 0..768 stored, 3 = full with the whole spectrum stored,
// 4 = magnitude class (MFCC + the whole spectrum stored + spectrum bands 26 / 27; kFramesStatsLater: the spectral
// statistics are taken from the stored magnitudes by bands_kernel, which runs for this mask anyway)
int frames32_class(uint32_t mask) {
  if (mask == 1u) return 0;
  if (!(mask & ~0xFFu)) return 1;
  if (mask & kFramesStatsLater) return 4;
  return (mask & kFramesWholeSpectrum) ? 3 : 2;
}

template <int FEAT, bool SCALED>
static hipError_t launch_frames32_class(const FrameArgs& a, int grid_blocks, hipStream_t stream) {
  constexpr int lds = Lds32<kMel32Rows>::total;
  auto k = frames32_kernel<FEAT, SCALED>;
  static bool attribute_set[16] = {};   // per device: raising the dynamic LDS limit once is enough
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attribute_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attribute_set[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(kWaves32 * 64), lds, stream, a);
  return hipGetLastError();
}

// f64 arithmetic, f32 PCM (scaled: the LoadSample front end's float signal times the buffer's FinalScaling): the
// MFCC-only class, or the statistics class (a.mask has bits 1..7, a.stat_tmp holds total_frames x 8 doubles) followed
// by its closed-form kernel
hipError_t launch_frames32(const FrameArgs& a, int grid_blocks, hipStream_t stream, int64_t total_frames, bool scaled) {
  if (a.n_chunks <= 0) return hipSuccess;
  const int cls = frames32_class(a.mask);
  if (cls == 0) return scaled ? launch_frames32_class<0, true>(a, grid_blocks, stream) : launch_frames32_class<0, false>(a, grid_blocks, stream);
  if (cls == 4) {
    // Class 6 (no upper spectrum) where nobody reads bins above 768 -- and where it was measured to pay: batches of 160 000
    // frames and more (C4-shaped files: -5 % at 160 k frames, -7 % from 320 k to 5.12 M).  The 82 000 frames of BASELINE's C3
    // (1 000 two-second files) run 15 % SLOWER on it and 80 000 frames of one-second files 3 % faster (not understood:
    // profiles/r06/ab_class6.txt); the stored values are the same either way, so the choice is free.
    constexpr int64_t kClass6Frames = 131072;
    if ((a.mask & kFramesWholeSpectrum) || total_frames < kClass6Frames)
      return scaled ? launch_frames32_class<4, true>(a, grid_blocks, stream) : launch_frames32_class<4, false>(a, grid_blocks, stream);
    return scaled ? launch_frames32_class<6, true>(a, grid_blocks, stream) : launch_frames32_class<6, false>(a, grid_blocks, stream);
  }
  hipError_t e;
  if (cls == 1 && !(a.mask & 0x30u))   // neither skewness nor kurtosis: no third / fourth moments
    e = scaled ? launch_frames32_class<5, true>(a, grid_blocks, stream) : launch_frames32_class<5, false>(a, grid_blocks, stream);
  else if (scaled)
    e = (cls == 1) ? launch_frames32_class<1, true>(a, grid_blocks, stream)
                   : (cls == 2 ? launch_frames32_class<2, true>(a, grid_blocks, stream) : launch_frames32_class<3, true>(a, grid_blocks, stream));
  else
    e = (cls == 1) ? launch_frames32_class<1, false>(a, grid_blocks, stream)
                   : (cls == 2 ? launch_frames32_class<2, false>(a, grid_blocks, stream) : launch_frames32_class<3, false>(a, grid_blocks, stream));
  if (e != hipSuccess) return e;
  return launch_stats32_finish(a, stream, total_frames);
}

hipError_t launch_stats32_finish(const FrameArgs& a, hipStream_t stream, int64_t total_frames) {
  if (total_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(stats32_finish_kernel, dim3((unsigned)((total_frames + 255) / 256)), dim3(256), 0, stream, a, total_frames);
  return hipGetLastError();
}

}  // namespace afx
