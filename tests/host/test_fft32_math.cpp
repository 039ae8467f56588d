// tests/host/test_fft32_math.cpp -- the in-register DFT building blocks of the gfx950 kernels on the host, against a direct
// DFT in long double (run by tests/test_host_cpp.py; no GPU, no library):
//   * afx_fft32.h (half-wave kernels, afx_frames32.hip / afx_rhythm.hip): radix4, bfly_tw, radix4_tw3 / tw4, dft16_rest
//     behind its first radix-4 stage, dft32_merge, dft32;
//   * afx_fft.h (64-lane kernels): cmul, radix4, dft16;
//   * the 1024-point transform composed the way frames32_kernel composes it -- 32 x 32: dft32 over n1 in "lane" n2, the
//     exchange, the factors w1024^(n2 k1) fused into the first radix-4 of the second pass (radix4_tw4), dft32_rest --
//     and the even / odd untangle of the 2048-point real transform that yields the bins the reference's 2048-point
//     complex transform of (x, 0) has (Fourier.cpp:243-270; OouraFFT8g.cpp:289, sign +: the magnitudes do not see it).
// The reference's own FFT test compares two implementations to 1e-4 (TestFourier.cpp:48-82); these blocks must meet
// 1e-14 of the largest output.
#include <cmath>
#include <algorithm>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../afec_amd/csrc/afx_fft.h"
#include "../../afec_amd/csrc/afx_fft32.h"

namespace {

using cld = std::complex<long double>;
const long double kPi = 3.14159265358979323846264338327950288L;

std::vector<cld> direct_dft(const std::vector<cld>& x) {
  const size_t n = x.size();
  std::vector<cld> X(n);
  for (size_t k = 0; k < n; ++k) {
    cld acc = 0;
    for (size_t j = 0; j < n; ++j) acc += x[j] * std::polar(1.0L, -2.0L * kPi * (long double)((j * k) % n) / (long double)n);
    X[k] = acc;
  }
  return X;
}

int g_failed = 0;
void expect_close(const char* what, const std::vector<cld>& want, const std::vector<cld>& got, long double tol) {
  long double top = 0, worst = 0;
  for (const cld& v : want) top = std::max(top, std::abs(v));
  for (size_t i = 0; i < want.size(); ++i) worst = std::max(worst, std::abs(want[i] - got[i]));
  const bool ok = worst <= tol * top;
  std::printf("%-58s max |err| / max |X| = %.3Le  %s\n", what, top > 0 ? worst / top : worst, ok ? "ok" : "FAILED");
  if (!ok) ++g_failed;
}

template <typename C>
std::vector<cld> to_ld(const C* v, int n) {
  std::vector<cld> r((size_t)n);
  for (int i = 0; i < n; ++i) r[(size_t)i] = cld(v[i].re, v[i].im);
  return r;
}

}  // namespace

int main() {
  std::mt19937_64 gen(2048);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  using c32 = afx::f32x32::cx<double>;
  using c64 = afx::cx<double>;

  for (int trial = 0; trial < 50; ++trial) {
    // ---- 4-point butterflies ----
    {
      c32 v[4]; c64 u[4];
      std::vector<cld> x(4);
      for (int i = 0; i < 4; ++i) { v[i] = {U(gen), U(gen)}; u[i] = {v[i].re, v[i].im}; x[(size_t)i] = cld(v[i].re, v[i].im); }
      afx::f32x32::radix4(v[0], v[1], v[2], v[3]);
      afx::radix4(u[0], u[1], u[2], u[3]);
      if (trial == 0) { expect_close("afx_fft32.h radix4", direct_dft(x), to_ld(v, 4), 1e-15L); expect_close("afx_fft.h radix4", direct_dft(x), to_ld(u, 4), 1e-15L); }
      // with factors: radix4 of (w0 a, w1 b, w2 c, w3 d)
      c32 w[4], t[4];
      std::vector<cld> y(4);
      for (int i = 0; i < 4; ++i) {
        const double th = U(gen) * 3.0;
        w[i] = {std::cos(th), -std::sin(th)};
        t[i] = {(double)x[(size_t)i].real(), (double)x[(size_t)i].imag()};
        y[(size_t)i] = x[(size_t)i] * cld(w[i].re, w[i].im);
      }
      c32 t3[4] = {t[0], t[1], t[2], t[3]};
      afx::f32x32::radix4_tw4(t[0], t[1], t[2], t[3], w[0], w[1], w[2], w[3]);
      if (trial == 0) expect_close("afx_fft32.h radix4_tw4", direct_dft(y), to_ld(t, 4), 4e-16L * 8);
      y[0] = x[0];
      afx::f32x32::radix4_tw3(t3[0], t3[1], t3[2], t3[3], w[1], w[2], w[3]);
      if (trial == 0) expect_close("afx_fft32.h radix4_tw3", direct_dft(y), to_ld(t3, 4), 4e-16L * 8);
      // x = e + w o, y = e - w o
      c32 e = {U(gen), U(gen)}, o = {U(gen), U(gen)};
      const double th = U(gen) * 3.0;
      const cld E(e.re, e.im), O(o.re, o.im), W = std::polar(1.0L, -(long double)th);
      afx::f32x32::bfly_tw(e, o, std::cos(th), std::sin(th));
      c32 eo[2] = {e, o};
      if (trial == 0) expect_close("afx_fft32.h bfly_tw", {E + W * O, E - W * O}, to_ld(eo, 2), 4e-16L * 8);
    }
    // ---- 16 points ----
    {
      c64 u[16]; c32 v[16];
      std::vector<cld> x(16);
      for (int i = 0; i < 16; ++i) { u[i] = {U(gen), U(gen)}; v[i] = {u[i].re, u[i].im}; x[(size_t)i] = cld(u[i].re, u[i].im); }
      afx::dft16(u);
      // dft16_rest expects the first radix-4 stage done: v[4c + b] = y[b][c]
      for (int b = 0; b < 4; ++b) afx::f32x32::radix4(v[b], v[4 + b], v[8 + b], v[12 + b]);
      afx::f32x32::dft16_rest(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
      if (trial < 2) { expect_close("afx_fft.h dft16", direct_dft(x), to_ld(u, 16), 1e-15L * 4); expect_close("afx_fft32.h radix4 + dft16_rest", direct_dft(x), to_ld(v, 16), 1e-15L * 4); }
      else { const auto want = direct_dft(x); long double top = 0, worst = 0; for (auto& q : want) top = std::max(top, std::abs(q));
             for (int i = 0; i < 16; ++i) worst = std::max({worst, std::abs(want[(size_t)i] - cld(u[i].re, u[i].im)), std::abs(want[(size_t)i] - cld(v[i].re, v[i].im))});
             if (worst > 4e-15L * top) { std::printf("16-point trial %d FAILED: %.3Le\n", trial, worst / top); ++g_failed; } }
    }
    // ---- 32 points ----
    {
      c32 v[32], m[32];
      std::vector<cld> x(32), xe(16), xo(16);
      for (int i = 0; i < 32; ++i) { v[i] = {U(gen), U(gen)}; x[(size_t)i] = cld(v[i].re, v[i].im); }
      afx::f32x32::dft32(v);
      const auto want = direct_dft(x);
      if (trial < 2) expect_close("afx_fft32.h dft32", want, to_ld(v, 32), 1e-15L * 6);
      else { long double top = 0, worst = 0; for (auto& q : want) top = std::max(top, std::abs(q));
             for (int i = 0; i < 32; ++i) worst = std::max(worst, std::abs(want[(size_t)i] - cld(v[i].re, v[i].im)));
             if (worst > 6e-15L * top) { std::printf("32-point trial %d FAILED: %.3Le\n", trial, worst / top); ++g_failed; } }
      // dft32_merge alone: v[2k] = E[k], v[2k+1] = O[k] of the even / odd inputs
      for (int k = 0; k < 16; ++k) { xe[(size_t)k] = x[(size_t)(2 * k)]; xo[(size_t)k] = x[(size_t)(2 * k + 1)]; }
      const auto E = direct_dft(xe), O = direct_dft(xo);
      for (int k = 0; k < 16; ++k) { m[2 * k] = {(double)E[(size_t)k].real(), (double)E[(size_t)k].imag()}; m[2 * k + 1] = {(double)O[(size_t)k].real(), (double)O[(size_t)k].imag()}; }
      afx::f32x32::dft32_merge(m);
      if (trial == 0) expect_close("afx_fft32.h dft32_merge", want, to_ld(m, 32), 1e-15L * 6);
    }
  }

  // ---- 1024 points as frames32_kernel composes them, and the real-input untangle ----
  for (int trial = 0; trial < 3; ++trial) {
    std::vector<double> xr(2048);
    for (double& s : xr) s = trial == 2 ? 0.0 : U(gen);
    if (trial == 2) xr[777] = 1.0;                                 // an impulse: flat spectrum
    if (trial == 1) for (int n = 0; n < 2048; ++n) xr[(size_t)n] = std::sin(2.0 * 3.14159265358979323846 * 46.0 * n / 2048.0);   // a bin-centred tone
    std::vector<cld> z(1024);
    for (int n = 0; n < 1024; ++n) z[(size_t)n] = cld(xr[(size_t)(2 * n)], xr[(size_t)(2 * n + 1)]);
    // pass 1: "lane" n2 holds z[n2 + 32 n1] in register n1
    std::vector<std::vector<c32>> Y(32, std::vector<c32>(32));     // Y[n2][k1]
    for (int n2 = 0; n2 < 32; ++n2) {
      c32 v[32];
      for (int n1 = 0; n1 < 32; ++n1) v[n1] = {(double)z[(size_t)(n2 + 32 * n1)].real(), (double)z[(size_t)(n2 + 32 * n1)].imag()};
      afx::f32x32::dft32(v);
      for (int k1 = 0; k1 < 32; ++k1) Y[(size_t)n2][(size_t)k1] = v[k1];
    }
    // exchange: "lane" k1 holds Y[n2][k1] in register n2; factors w1024^(n2 k1) fused into the first radix-4 of pass 2
    std::vector<cld> Z(1024);
    for (int k1 = 0; k1 < 32; ++k1) {
      c32 v[32], w[32];
      for (int n2 = 0; n2 < 32; ++n2) {
        v[n2] = Y[(size_t)n2][(size_t)k1];
        const long double th = -2.0L * kPi * (long double)(n2 * k1) / 1024.0L;
        w[n2] = {(double)std::cos(th), (double)std::sin(th)};
      }
      for (int j = 0; j < 8; ++j) afx::f32x32::radix4_tw4(v[j], v[j + 8], v[j + 16], v[j + 24], w[j], w[j + 8], w[j + 16], w[j + 24]);
      afx::f32x32::dft32_rest(v);
      for (int k2 = 0; k2 < 32; ++k2) Z[(size_t)(k1 + 32 * k2)] = cld(v[k2].re, v[k2].im);
    }
    expect_close(trial == 0 ? "1024 points, 32 x 32 (noise)" : trial == 1 ? "1024 points, 32 x 32 (tone)" : "1024 points, 32 x 32 (impulse)", direct_dft(z), Z, 1e-14L);
    // untangle: X[k] = E[k] + w2048^k O[k], E = (Z[k] + conj Z[N-k]) / 2, O = (Z[k] - conj Z[N-k]) / (2 i), k = 0..1023
    std::vector<cld> x2048(2048), X(1024);
    for (int n = 0; n < 2048; ++n) x2048[(size_t)n] = cld(xr[(size_t)n], 0);
    const auto want = direct_dft(x2048);
    for (int k = 0; k < 1024; ++k) {
      const cld a = Z[(size_t)k], b = std::conj(Z[(size_t)((1024 - k) % 1024)]);
      const cld E = (a + b) * 0.5L, O = (a - b) * cld(0, -0.5L);
      X[(size_t)k] = E + std::polar(1.0L, -2.0L * kPi * (long double)k / 2048.0L) * O;
    }
    expect_close("2048-point real transform by the untangle, bins 0..1023", std::vector<cld>(want.begin(), want.begin() + 1024), X, 1e-14L);
  }
  std::printf(g_failed ? "test_fft32_math: %d FAILED\n" : "test_fft32_math: all passed\n", g_failed);
  return g_failed ? 1 : 0;
}
