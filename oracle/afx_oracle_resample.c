/* oracle/afx_oracle_resample.c -- TEST INFRASTRUCTURE ONLY (see afx_oracle.h): the sample-rate conversion of
 * TSampleAnalyser::LoadSample (SampleAnalyser.cpp:563-607), i.e. libresample 0.1.3 as the reference drives it:
 *
 *   resample_open(highQuality = 1, factor, factor); resample_process(h, factor, in, n, lastFlag = 1, &used, out, NewSize)
 *   with factor = 1 / Speed, Speed = (double)file_rate / (double)analyser_rate, NewSize = max(1, d2iRound(n / Speed))
 *
 * restated without the library's ring buffers: libresample copies the input through a 4096-sample window X (with
 * Xoff zeros in front of the first and behind the last sample) and keeps the converter's time relative to that
 * window; here the window is only its bookkeeping (where it starts in the input, `base`), every output sample reads
 * the input directly, and samples outside [0, n) are the zeros the library pads with.  What must be, and is,
 * reproduced operation by operation: the time recurrence (CurrentTime += dt in double, the per-window
 * "Time -= Nx", the creep correction), the filter index arithmetic, and the float accumulation order of the two
 * wings.  interpFilt is FALSE in resample_process (resample.c:171): the coefficient deltas are never used.
 *
 *   filter table        3rdParty/Resample/Dist/src/filterkit.c:66-113 (lrsLpFilter, Izero), resample.c:110-131
 *   window bookkeeping  resample.c:133-150 (Xoff, XSize), 214-330 (resample_process)
 *   time loop           resamplesubs.c:30-66 (lrsSrcUp), 70-118 (lrsSrcUD)
 *   wings               filterkit.c:115-170 (lrsFilterUp), 172-215 (lrsFilterUD)
 *
 * PINNED against oracle/_ref/ref_driver `resample` (the reference's own libresample sources, compiled where they
 * lie): tests/golden/resample.npz, bit-exact (tests/test_oracle_resample.py). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "afx_oracle.h"

#define RS_NPC 4096   /* resample_defs.h:72 */
#define RS_NMULT 35   /* resample.c:104 (highQuality) */
#define RS_NWING (RS_NPC * (RS_NMULT - 1) / 2)

static double rs_izero(double x) {   /* filterkit.c:66-81 */
  double sum, u, halfx, temp;
  int n;
  sum = u = n = 1;
  halfx = x / 2.0;
  do {
    temp = halfx / (double)n;
    n += 1;
    temp *= temp;
    u *= temp;
    sum += u;
  } while (u >= 1E-21 * sum);
  return sum;
}

/* Imp[RS_NWING]: the right wing of the Kaiser-windowed sinc as floats (resample.c:110-124) */
void afx_oracle_resample_filter(float* imp) {
  const double pi = 3.14159265358979232846;   /* resample_defs.h:31 */
  const double frq = 0.5 * 0.90, beta = 6;    /* resample.c:113-119 */
  const int n = RS_NWING;
  double* c = (double*)malloc(sizeof(double) * (size_t)n);
  double ibeta, inm1;
  int i;
  c[0] = 2.0 * frq;
  for (i = 1; i < n; i++) {
    const double temp = pi * (double)i / (double)RS_NPC;
    c[i] = sin(2.0 * temp * frq) / temp;
  }
  ibeta = 1.0 / rs_izero(beta);
  inm1 = 1.0 / ((double)(n - 1));
  for (i = 1; i < n; i++) {
    double temp = (double)i * inm1;
    double temp1 = 1.0 - temp * temp;
    temp1 = (temp1 < 0 ? 0 : temp1);
    c[i] *= rs_izero(beta * sqrt(temp1)) * ibeta;
  }
  for (i = 0; i < n; i++) imp[i] = (float)c[i];
  free(c);
}

static float rs_at(const float* in, int64_t n, int64_t i) { return (i >= 0 && i < n) ? in[i] : 0.0f; }

/* one output sample at converter time `t` of a window that starts at input sample `base` */
static float rs_output(const float* imp, const float* in, int64_t n, int64_t base, double t, double factor, float lpscl) {
  const double left_phase = t - floor(t), right_phase = 1.0 - left_phase;
  const int64_t c = base + (int64_t)(int)t;
  float v, w;
  int64_t k;
  if (factor >= 1) {   /* lrsFilterUp, filterkit.c:115-170 */
    double ph = left_phase * RS_NPC;
    int h = (int)ph;
    v = 0.0f;
    for (k = 0; h < RS_NWING; h += RS_NPC, ++k) { float t1 = imp[h]; t1 *= rs_at(in, n, c - k); v += t1; }
    ph = right_phase * RS_NPC;
    h = (int)ph;
    if (ph == 0) h += RS_NPC;
    w = 0.0f;
    for (k = 0; h < RS_NWING - 1; h += RS_NPC, ++k) { float t1 = imp[h]; t1 *= rs_at(in, n, c + 1 + k); w += t1; }
  } else {             /* lrsFilterUD, filterkit.c:172-215 */
    const double dh = fmin((double)RS_NPC, factor * RS_NPC);   /* resamplesubs.c:91 */
    double ho = left_phase * dh;
    v = 0.0f;
    for (k = 0; (int)ho < RS_NWING; ho += dh, ++k) { float t1 = imp[(int)ho]; t1 *= rs_at(in, n, c - k); v += t1; }
    ho = right_phase * dh;
    if (right_phase == 0) ho += dh;
    w = 0.0f;
    for (k = 0; (int)ho < RS_NWING - 1; ho += dh, ++k) { float t1 = imp[(int)ho]; t1 *= rs_at(in, n, c + 1 + k); w += t1; }
  }
  v += w;
  v *= lpscl;
  return v;
}

/* -> malloc'ed out[*n_out] (afx_oracle_free); *n_written = samples the converter produced (the reference leaves the
 * rest of its buffer uninitialised and asserts that there is none, SA:596-597; here it is 0) */
float* afx_oracle_resample(const float* in, int64_t n, int file_rate, int analyser_rate, int64_t* n_out, int64_t* n_written) {
  const double speed = (double)file_rate / (double)analyser_rate;   /* SA:563 */
  const double factor = 1.0 / speed;
  const double dt = 1.0 / factor;                                   /* resamplesubs.c:45 */
  const double rounded = (double)n / speed;
  int64_t new_size = (int64_t)(int)(rounded + (rounded > 0 ? 0.5 : (rounded < 0 ? -0.5 : 0.0)));   /* TMath::d2iRound, InlineMath.inl:823-826 */
  float lpscl = 1.0f;
  float* imp = (float*)malloc(sizeof(float) * RS_NWING);
  float* out;
  unsigned xoff, xsize, xread;
  int64_t used = 0, base, written = 0;
  double time;
  if (new_size < 1) new_size = 1;
  out = (float*)calloc((size_t)new_size, sizeof(float));
  afx_oracle_resample_filter(imp);
  if (factor < 1) lpscl = lpscl * factor;                           /* resample.c:211-212 */
  {
    const double reach = ((RS_NMULT + 1) / 2.0) * fmax(1.0, 1.0 / factor) + 10;   /* resample.c:133-135 */
    xoff = (unsigned)reach;
    xsize = (2 * xoff + 10 > 4096) ? 2 * xoff + 10 : 4096;                         /* resample.c:142 */
  }
  xread = xoff;
  time = (double)xoff;
  base = -(int64_t)xoff;
  for (;;) {
    int64_t len = (int64_t)xsize - xread;
    int nx, ncreep;
    unsigned xp;
    double t, end_time;
    if (len >= n - used) len = n - used;
    used += len;
    xread += (unsigned)len;
    nx = (used == n) ? (int)xread - (int)xoff : (int)xread - 2 * (int)xoff;   /* resample.c:240-250 */
    if (nx <= 0) break;
    t = time;
    end_time = t + nx;
    while (t < end_time) {                                                      /* resamplesubs.c:49-62 */
      if (written < new_size) out[written] = rs_output(imp, in, n, base, t, factor, lpscl);
      ++written;
      t += dt;
    }
    time = t;
    time -= nx;                                                                 /* resample.c:271-280 */
    xp = xoff + (unsigned)nx;
    ncreep = (int)time - (int)xoff;
    if (ncreep) { time -= ncreep; xp += (unsigned)ncreep; }
    base += (int64_t)xp - xoff;                                                 /* resample.c:283-292: the window moves on */
    xread = xread - (xp - xoff);
    if (written >= new_size) break;                                             /* resample.c:305-318: the output buffer is full */
  }
  free(imp);
  *n_out = new_size;
  *n_written = written < new_size ? written : new_size;
  return out;
}
