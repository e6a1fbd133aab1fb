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

/*
 * x / c for a small integer constant c, bit-identical to the IEEE division the
 * reference performs (out[k] /= smooth_width, ... / 10): q0 = x * RN(1/c) is within
 * 2 ulp, the remainder fma is exact and one correction step lands on the correctly
 * rounded quotient (x/c is never within 2^-100 of a rounding boundary unless it is
 * exactly representable).  Three multiply-class instructions instead of the ~30 of a
 * software f64 division — this sits on the per-step critical path of the tail.
 * Checked against `/` on the CPU by tests/host/test_tail_host.cpp.
 */
BL_THD double bl_div_const(double x, double c, double rc) {
  const double q0 = x * rc;
  const double r0 = __builtin_fma(-q0, c, x);
  return __builtin_fma(r0, rc, q0);
}
#define BL_DIV19(x) bl_div_const((x), 19.0, 1.0 / 19.0)
#define BL_DIV10(x) bl_div_const((x), 10.0, 1.0 / 10.0)

/* peak detector over the twice-smoothed signal (ref :275-280) */
struct bl_peaks {
  double p1, p2; /* out2[i-1], out2[i-2] */
  int beat, i;   /* i: index of the next value */
  BL_THD void init() { p1 = 0; p2 = 0; beat = 0; i = 0; }
  BL_THD void push(double v) {
    if (i >= 2) {
      const float epsilon = 0.000001f;
      double dl = p1 - p2, dr = p1 - v;
      if (dl > epsilon && dr > epsilon) beat++;
    }
    p2 = p1;
    p1 = v;
    ++i;
  }
};

/*
 * One bl_rectangular_filter(out, in, N, 19) as a stream.  push(i, in_i, old_i)
 * must be called for i = 0..N-1 in order, then finish(); old_i is the previous
 * content of out[i].  Outputs are handed to `sink.push(out_i)` in index order:
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
  int N, t, s19, s10; /* next input index and its ring slots (t % 19, t % 10) */
  BL_THD void init(int n) { run = 0; N = n; t = 0; s19 = 0; s10 = 0; }

  template <typename SINK>
  BL_THD void push(double v, double old, double *ring, double *olds, int stride, SINK &sink) {
    if (t < BL_BOX) {
      run += v; /* ref :25-26 */
    } else {
      run -= ring[s19 * stride]; /* in[t-19], ref :30 */
      run += v;                  /* ref :31 */
    }
    ring[s19 * stride] = v;
    if (KEEP_OLD) olds[s10 * stride] = old;
    if (t <= 8) {
      sink.push(BL_DIV19(old));
    } else if (t >= BL_BOX - 1 && t <= N - 2) {
      sink.push(BL_DIV19(run)); /* out[k + half - 1] = tempsum, k = t-18 */
    }
    ++t;
    s19 = s19 == BL_BOX - 1 ? 0 : s19 + 1;
    s10 = s10 == 9 ? 0 : s10 + 1;
  }

  template <typename SINK>
  BL_THD void finish(const double *ring, const double *olds, int stride, SINK &sink) {
    /* out[N - half] += in[k], k = N-19 .. N-1 (ref :34-35) */
    double acc = KEEP_OLD ? olds[((N - 10) % 10) * stride] : 0.0;
    for (int k = N - BL_BOX; k < N; ++k) acc += ring[(k % BL_BOX) * stride];
    sink.push(BL_DIV19(acc));
    for (int i = N - 9; i < N; ++i) {
      double old = KEEP_OLD ? olds[(i % 10) * stride] : 0.0;
      sink.push(BL_DIV19(old));
    }
  }
};

/* second box filter feeding the peak detector */
struct bl_tail_stage2 {
  bl_box19<false> box;
  bl_peaks peaks;
  double *ring;
  int stride;
  BL_THD void push(double v) { box.push(v, 0.0, ring, (double *)0, stride, peaks); }
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
    const double wa = (1 - lambda) * y + BL_DIV10(lambda * 172 * dj); /* ref :230-231 */
    x6 = x5; x5 = x4; x4 = x3; x3 = x2; x2 = x1; x1 = x;
    y6 = y5; y5 = y4; y4 = y3; y3 = y2; y2 = y1; y1 = y;
    double ss = 0; /* ref :259-263: smoothed_sum[N-1] stays 0 */
    if (j <= N - 2) { atk += wa; ss = wa; } /* ref :246-248 */
    box1.push(ss, wa, ring1, olds1, stride, st2);
  }

  BL_THD void finish() {
    box1.finish(ring1, olds1, stride, st2);
    st2.box.finish(st2.ring, (const double *)0, stride, st2.peaks);
  }

  BL_THD int beat() const { return st2.peaks.beat; }
};

/*
 * The same stream cut in three for three cooperating waves (k_env_tail): the recurrence
 * (bl_tail_iir: a chain of 8 dependent operations per step), the onset weighting and first box
 * filter (bl_tail_ab) and the second box filter with the peak test (bl_tail_c).  One wave per 64
 * songs issuing everything in order needed ~340 cycles per step, two waves (recurrence |
 * the rest) ~170; with three the recurrence's own chain (~90 cycles) is what is left.  bl_tail above
 * is the plain one-step-at-a-time form; the three stages produce bit for bit the same
 * (tests/host/test_tail_host.cpp runs both against the oracle).
 */
struct bl_tail_iir {
  double x2, x4, x6;             /* even inputs x[j-2], x[j-4], x[j-6]; odd ones are stuffed zeros */
  double y1, y2, y3, y4, y5, y6; /* y[j-1..j-6] */
  BL_THD void init() { x2 = x4 = x6 = 0; y1 = y2 = y3 = y4 = y5 = y6 = 0; }
  BL_THD double feedback(double d) {
    double c = 0; /* ref :214-215 */
    c += BL_BUT_A1 * y1;
    c += BL_BUT_A2 * y2;
    c += BL_BUT_A3 * y3;
    c += BL_BUT_A4 * y4;
    c += BL_BUT_A5 * y5;
    c += BL_BUT_A6 * y6;
    const double y = (d - c) / BL_BUT_A0; /* ref :216 */
    y6 = y5; y5 = y4; y4 = y3; y3 = y2; y2 = y1; y1 = y;
    return y;
  }
  /* steps j (even, input x >= 0) and j + 1 (stuffed zero), any even j >= 0.  The feed-forward
   * sums of ref :210-213 skip their `+= b[k] * 0.0` terms: every partial sum is a sum of
   * products of positive taps with non-negative inputs, so adding +0.0 is the identity (and
   * 0 + b0*x is b0*x). */
  BL_THD void pair(double x, double &y_even, double &y_odd) {
    double d = BL_BUT_B0 * x;
    d += BL_BUT_B2 * x2;
    d += BL_BUT_B2 * x4;
    d += BL_BUT_B0 * x6;
    y_even = feedback(d);
    double e = BL_BUT_B1 * x;
    e += BL_BUT_B3 * x2;
    e += BL_BUT_B1 * x4;
    y_odd = feedback(e);
    x6 = x4; x4 = x2; x2 = x;
  }
};

/* A stream of doubles in LDS, written by one wave and read by the next: slot q of a lane is
 * base[q * stride] */
struct bl_tail_fifo {
  double *base;
  int stride, count;
  BL_THD void push(double v) {
    base[count * stride] = v;
    ++count;
  }
};

/*
 * Stage "AB": y_j -> onset difference, weighting, atk, first box filter -> the stream o1 of box-1
 * outputs (ref :221-263, :267-268).  Its outputs go to a sink (a bl_tail_fifo in the kernel, the
 * next stage directly in the host test): one per step in the steady state, several at the edges
 * (see bl_box19).
 */
struct bl_tail_ab {
  double yp; /* y[j-1] */
  double atk;
  bl_box19<true> box1;
  double *ring1, *olds1;
  int stride, N;

  /* scratch: 19 + 10 doubles per song, element e at scratch[e * stride] */
  BL_THD void init(int nb_frames, double *scratch, int stride_) {
    yp = 0;
    atk = 0;
    N = 2 * nb_frames;
    stride = stride_;
    ring1 = scratch;
    olds1 = scratch + 19 * stride_;
    box1.init(N);
  }

  /* Steady state: for 40 <= j and j + 1 <= N - 12 every conditional of step() is decided and box 1
   * emits exactly one value per step.  A block of 38 = 2 * 19 steps is two full turns of the ring,
   * so it can live in registers and come back to the slots it left; at least 11 generic steps
   * follow any such block, which refill the 10 `old` cells that the block does not maintain. */
  BL_THD static bool steady(int j, int n) { return (j & 1) == 0 && j >= 40 && j + 1 <= n - 12; }
  BL_THD static bool chunk_ok(int j, int n) { return steady(j, n) && steady(j + 36, n); }

  BL_THD double weighted(double y, double dj) {
    const float lambda = 0.8f; /* ref :171 */
    return (1 - lambda) * y + BL_DIV10(lambda * 172 * dj); /* ref :230-231 */
  }

  /* one step j = 0..N-1 from y_j (ref :221-263) */
  template <typename SINK> BL_THD void step(int j, double y, SINK &sink) {
    double dj;
    if (j == 0) dj = y;
    else { dj = y - yp; dj = dj > 0 ? dj : 0; }
    const double wa = weighted(y, dj);
    yp = y;
    double ss = 0;
    if (j <= N - 2) { atk += wa; ss = wa; }
    box1.push(ss, wa, ring1, olds1, stride, sink);
  }
  template <typename SINK> BL_THD void finish(SINK &sink) { box1.finish(ring1, olds1, stride, sink); }

  /* One steady step with the ring in registers (ra: the 19 most recent inputs of box 1 in ring
   * order, P the static slot), as two stages with no arithmetic in common — A: y_s -> wa_s, atk;
   * B: wa_s -> o1_s.  A GPU wave issues in order and is alone on its SIMD here: written step after
   * step the operations of a step form one dependent chain.  reg_pipe() runs A of step s + 1
   * beside B of step s: the same operations on the same operands, two chains in flight. */
  template <int P> BL_THD double stage_a(double y) {
    double dj = y - yp;
    dj = dj > 0 ? dj : 0;
    const double wa = weighted(y, dj);
    yp = y;
    atk += wa;
    return wa;
  }
  template <int P> BL_THD double stage_b(double wa, double (&ra)[BL_BOX]) {
    box1.run -= ra[P];
    box1.run += wa;
    ra[P] = wa;
    return BL_DIV19(box1.run);
  }
  template <int T>
  BL_THD void reg_pipe(const double *yin, int ystride, double *o1out, int ostride, double (&ra)[BL_BOX],
                       double (&wa)[2]) {
    if constexpr (T < 2 * BL_BOX + 1) {
      if constexpr (T >= 1) o1out[(T - 1) * ostride] = stage_b<(T - 1) % BL_BOX>(wa[(T - 1) % 2], ra);
      if constexpr (T < 2 * BL_BOX) wa[T % 2] = stage_a<T % BL_BOX>(yin[T * ystride]);
      reg_pipe<T + 1>(yin, ystride, o1out, ostride, ra, wa);
    }
  }

  /* 38 steady steps y_j .. y_(j+37) at yin[s * ystride] -> 38 box-1 outputs at o1out[s * ostride];
   * requires chunk_ok(j, N) */
  BL_THD void fast_chunk38(const double *yin, int ystride, double *o1out, int ostride) {
    double ra[BL_BOX];
    {
      int sa = box1.s19;
#pragma unroll
      for (int k = 0; k < BL_BOX; ++k) {
        ra[k] = ring1[sa * stride];
        sa = sa == BL_BOX - 1 ? 0 : sa + 1;
      }
    }
    double wa[2];
    reg_pipe<0>(yin, ystride, o1out, ostride, ra, wa);
    {
      int sa = box1.s19;
#pragma unroll
      for (int k = 0; k < BL_BOX; ++k) {
        ring1[sa * stride] = ra[k];
        sa = sa == BL_BOX - 1 ? 0 : sa + 1;
      }
    }
    box1.t += 38;
    box1.s10 = (box1.s10 + 8) % 10;
  }
};

/*
 * Stage "C": the stream o1 -> second box filter -> peak test (ref :269-280).  push() takes the
 * box-1 outputs in index order; after the N-th one finish() flushes the filter's tail.
 */
struct bl_tail_c {
  bl_box19<false> box;
  bl_peaks peaks;
  double *ring;
  int stride, N, taken; /* taken: box-1 outputs consumed so far */

  /* scratch: 19 doubles per song */
  BL_THD void init(int nb_frames, double *scratch, int stride_) {
    N = 2 * nb_frames;
    stride = stride_;
    ring = scratch;
    taken = 0;
    box.init(N);
    peaks.init();
  }
  BL_THD void push(double v) {
    box.push(v, 0.0, ring, (double *)0, stride, peaks);
    ++taken;
  }
  BL_THD void finish() { box.finish(ring, (const double *)0, stride, peaks); }
  BL_THD int beat() const { return peaks.beat; }

  /* 38 more inputs keep box 2 in its steady state (ring full, one output per input, ref :28-32) and
   * the peak detector past its first two values */
  BL_THD bool chunk_ok() const { return taken >= BL_BOX && taken + 37 <= N - 2; }

  template <int P> BL_THD void stage_c(double o1, double (&rb)[BL_BOX]) {
    box.run -= rb[P];
    box.run += o1;
    rb[P] = o1;
    const double o2 = BL_DIV19(box.run);
    const float epsilon = 0.000001f;
    const double dl = peaks.p1 - peaks.p2, dr = peaks.p1 - o2;
    peaks.beat += (dl > epsilon && dr > epsilon) ? 1 : 0;
    peaks.p2 = peaks.p1;
    peaks.p1 = o2;
  }
  template <int T> BL_THD void reg_steps(const double *o1in, int istride, double (&rb)[BL_BOX]) {
    if constexpr (T < 2 * BL_BOX) {
      stage_c<T % BL_BOX>(o1in[T * istride], rb);
      reg_steps<T + 1>(o1in, istride, rb);
    }
  }
  /* 38 steady inputs at o1in[s * istride]; requires chunk_ok() */
  BL_THD void fast_chunk38(const double *o1in, int istride) {
    double rb[BL_BOX];
    {
      int sb = box.s19;
#pragma unroll
      for (int k = 0; k < BL_BOX; ++k) {
        rb[k] = ring[sb * stride];
        sb = sb == BL_BOX - 1 ? 0 : sb + 1;
      }
    }
    reg_steps<0>(o1in, istride, rb);
    {
      int sb = box.s19;
#pragma unroll
      for (int k = 0; k < BL_BOX; ++k) {
        ring[sb * stride] = rb[k];
        sb = sb == BL_BOX - 1 ? 0 : sb + 1;
      }
    }
    box.t += 38;
    box.s10 = (box.s10 + 8) % 10;
    peaks.i += 38;
    taken += 38;
  }
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
