/*
 * bl_tail.h — streaming form of parts 2 and 3 of bl_envelope_sort
 * (ref src/tempo_atk_sort.c:184-284 and bl_rectangular_filter, :19-40).
 *
 * The reference materialises five arrays of 2*nb_frames doubles; here the same
 * arithmetic (same operands, same order, no contraction) runs as one pass with
 * O(1) state per song so that one GPU lane can own one song:
 *   x_j   = log(1 + mu*f[j/2]) / log(1 + mu)   (j even), 0 (j odd)      :186-190
 *   y_j   = 6th-order Butterworth recurrence                            :201-218
 *   d_j   = y_0 (j = 0) | max(y_j - y_(j-1), 0)                         :221-226
 *   wa_j  = (1-lambda)*y_j + lambda*172*d_j/10                          :229-232
 *   atk  += wa_j (j <= N-2)                                             :246-248
 *   box filter 19 twice (second input = first output, edge cells keep the
 *   previous contents of the destination array divided by 19)          :267-270
 *   beat  = #{ j in [1, N-2] : local maximum by more than 1e-6f }      :277-280
 * The two box filters are push-driven streams (bl_box19); see the emission
 * schedule in the comments of push().
 *
 * __host__ __device__ so that tests/host/test_tail_host.cpp can run the very
 * same code against the oracle on the CPU.  Compile with -ffp-contract=off.
 */
#ifndef BL_TAIL_H_
#define BL_TAIL_H_

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BL_THD __host__ __device__ __forceinline__
#else
#include <math.h>
#define BL_THD inline
#endif

#define BL_BOX 19
#define BL_BOX_HALF 10 /* (int)round(19/2.) — ref src/tempo_atk_sort.c:22 */

/* Butterworth low-pass: literal digits of ref include/bandpass_coeffs.h:484-492 */
#define BL_BUT_B0 1.9510e-05
#define BL_BUT_B1 1.1706e-04
#define BL_BUT_B2 2.9266e-04
#define BL_BUT_B3 3.9021e-04
#define BL_BUT_A0 1.00000
#define BL_BUT_A1 (-4.59007)
#define BL_BUT_A2 8.91034
#define BL_BUT_A3 (-9.34191)
#define BL_BUT_A4 5.56998
#define BL_BUT_A5 (-1.78845)
#define BL_BUT_A6 0.24136

/* peak detector over the twice-smoothed signal (ref :275-280) */
struct bl_peaks {
  double p1, p2; /* out2[i-1], out2[i-2] */
  int beat;
  BL_THD void init() { p1 = 0; p2 = 0; beat = 0; }
  BL_THD void push(int i, double v) {
    if (i >= 2) {
      const float epsilon = 0.000001f;
      double dl = p1 - p2, dr = p1 - v;
      if (dl > epsilon && dr > epsilon) beat++;
    }
    p2 = p1;
    p1 = v;
  }
};

/*
 * One bl_rectangular_filter(out, in, N, 19) as a stream.  push(i, in_i, old_i)
 * must be called for i = 0..N-1 in order, then finish(); old_i is the previous
 * content of out[i].  Outputs are handed to `sink.push(i, out_i)` in order:
 *   i <= 8          at push(i)       : old_i / 19
 *   9 <= i <= N-11  at push(i + 9)   : R_(i-9) / 19, R_0 = in_0+..+in_18,
 *                                      R_k = (R_(k-1) - in_(k-1)) + in_(k+18)
 *   i = N-10        at finish()      : (old + in_(N-19) + .. + in_(N-1)) / 19
 *   i >= N-9        at finish()      : old_i / 19
 * ring: 19 slots, element s at ring[s * stride]; olds: the last 10 old values
 * (10 slots, same stride) — only touched when KEEP_OLD.
 */
template <bool KEEP_OLD> struct bl_box19 {
  double run;
  int N;
  BL_THD void init(int n) { run = 0; N = n; }

  template <typename SINK>
  BL_THD void push(int t, double v, double old, double *ring, double *olds, int stride,
                   SINK &sink) {
    const int slot = t % BL_BOX;
    if (t < BL_BOX) {
      run += v; /* ref :25-26 */
    } else {
      run -= ring[slot * stride]; /* in[t-19], ref :30 */
      run += v;                   /* ref :31 */
    }
    ring[slot * stride] = v;
    if (KEEP_OLD) olds[(t % 10) * stride] = old;
    if (t <= 8) {
      sink.push(t, old / BL_BOX);
    } else if (t >= BL_BOX - 1 && t <= N - 2) {
      sink.push(t - 9, run / BL_BOX); /* out[k + half - 1] = tempsum, k = t-18 */
    }
  }

  template <typename SINK>
  BL_THD void finish(const double *ring, const double *olds, int stride, SINK &sink) {
    /* out[N - half] += in[k], k = N-19 .. N-1 (ref :34-35) */
    double acc = KEEP_OLD ? olds[((N - 10) % 10) * stride] : 0.0;
    for (int k = N - BL_BOX; k < N; ++k) acc += ring[(k % BL_BOX) * stride];
    sink.push(N - BL_BOX_HALF, acc / BL_BOX);
    for (int i = N - 9; i < N; ++i) {
      double old = KEEP_OLD ? olds[(i % 10) * stride] : 0.0;
      sink.push(i, old / BL_BOX);
    }
  }
};

/* second box filter feeding the peak detector */
struct bl_tail_stage2 {
  bl_box19<false> box;
  bl_peaks peaks;
  double *ring;
  int stride;
  BL_THD void push(int i, double v) { box.push(i, v, 0.0, ring, (double *)0, stride, peaks); }
};

struct bl_tail {
  double x1, x2, x3, x4, x5, x6; /* IIR input history  x[j-1..j-6] */
  double y1, y2, y3, y4, y5, y6; /* IIR output history y[j-1..j-6] */
  double atk;
  bl_box19<true> box1;
  bl_tail_stage2 st2;
  double *ring1, *olds1;
  int stride, N;

  /* scratch: 19 + 10 + 19 doubles per song, element e at scratch[e * stride] */
  BL_THD void init(int nb_frames, double *scratch, int stride_) {
    x1 = x2 = x3 = x4 = x5 = x6 = 0;
    y1 = y2 = y3 = y4 = y5 = y6 = 0;
    atk = 0;
    N = 2 * nb_frames;
    stride = stride_;
    ring1 = scratch;
    olds1 = scratch + 19 * stride_;
    st2.ring = scratch + 29 * stride_;
    st2.stride = stride_;
    box1.init(N);
    st2.box.init(N);
    st2.peaks.init();
  }

  /* one step j = 0..N-1; x is the (log-compressed, zero-stuffed) input sample */
  BL_THD void step(int j, double x) {
    /* d: ascending k, starting from 0 (ref :210-213); b[k] symmetric */
    double d = 0;
    d += BL_BUT_B0 * x;
    d += BL_BUT_B1 * x1;
    d += BL_BUT_B2 * x2;
    d += BL_BUT_B3 * x3;
    d += BL_BUT_B2 * x4;
    d += BL_BUT_B1 * x5;
    d += BL_BUT_B0 * x6;
    double c = 0; /* ref :214-215 */
    c += BL_BUT_A1 * y1;
    c += BL_BUT_A2 * y2;
    c += BL_BUT_A3 * y3;
    c += BL_BUT_A4 * y4;
    c += BL_BUT_A5 * y5;
    c += BL_BUT_A6 * y6;
    const double y = (d - c) / BL_BUT_A0; /* ref :216 */
    double dj; /* ref :221-226 */
    if (j == 0) dj = y;
    else { dj = y - y1; dj = dj > 0 ? dj : 0; }
    const float lambda = 0.8f; /* ref :171 */
    const double wa = (1 - lambda) * y + lambda * 172 * dj / 10; /* ref :230-231 */
    x6 = x5; x5 = x4; x4 = x3; x3 = x2; x2 = x1; x1 = x;
    y6 = y5; y5 = y4; y4 = y3; y3 = y2; y2 = y1; y1 = y;
    double ss = 0; /* ref :259-263: smoothed_sum[N-1] stays 0 */
    if (j <= N - 2) { atk += wa; ss = wa; } /* ref :246-248 */
    box1.push(j, ss, wa, ring1, olds1, stride, st2);
  }

  BL_THD void finish() {
    box1.finish(ring1, olds1, stride, st2);
    st2.box.finish(st2.ring, (const double *)0, stride, st2.peaks);
  }

  BL_THD int beat() const { return st2.peaks.beat; }
};

/* ref src/tempo_atk_sort.c:186-188 with mu = 100.0f; log101 = log(1 + mu) */
BL_THD double bl_tail_compress(double f, double log101) {
  const float mu = 100.0f;
  return log(1 + mu * f) / log101;
}

/* ref src/tempo_atk_sort.c:283-284 */
BL_THD float bl_tail_tempo(int beat, unsigned long long duration) {
  return (float)(4 * (float)beat / (float)duration - 30.4);
}
BL_THD float bl_tail_attack(double atk_sum, int n_samples) {
  return (float)(-1.74 * atk_sum * 10000 / n_samples + 58.3);
}

#endif /* BL_TAIL_H_ */
