// Diagnostics of the convolution kernels (csrc/conv_igemm.hip, csrc/conv_rows.hip), all in one place so that the shipped
// hot loops read clean.  A normal build defines none of the switches below: every macro here then expands to nothing (or to
// "true") and the kernels contain no diagnostic instruction.
//
//   -DSMOS_CONV_ABLATE=<bits>  tools/ablate_conv.sh: removes one ingredient of the stage at a time (1 barrier, 2 activation
//                              requests, 4 weight ring traffic, 8 epilogue stores) to time what is left; results are wrong.
//   -DSMOS_CONV_SCHED=0        the coarser cut of the stage (four MFMA groups) the in-kernel stamps were written for; the
//                              shipped cut (1) is eight half groups.
//   -DSMOS_CONV_STAMPS         tools/conv_stamps.py: cycles a wave spends in each segment of the stage (s_memtime), summed over
//                              its stages and written to a buffer of their own; needs SMOS_CONV_SCHED=0.
#pragma once

#ifndef SMOS_CONV_ABLATE
#define SMOS_CONV_ABLATE 0
#endif
#define SMOS_CONV_KEEPS(bit) (!(SMOS_CONV_ABLATE & (bit)))
#ifndef SMOS_CONV_SCHED
#define SMOS_CONV_SCHED 1
#endif

#ifdef SMOS_CONV_STAMPS
#define SMOS_STAMPS_ARG unsigned long long* stamps;
#define SMOS_STAMPS_DECLARE() unsigned long long stamp_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0
#define SMOS_STAMP(k)                                                                              \
  do {                                                                                             \
    unsigned long long t_;                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    stamp_sum[k] += t_ - stamp_last;                                                               \
    stamp_last = t_;                                                                               \
  } while (0)
#define SMOS_STAMPS_BEGIN()                                                                 \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");        \
  const unsigned long long clk0 = stamp_last, real0 = __builtin_amdgcn_s_memrealtime()
#define SMOS_STAMPS_END()                                                                                      \
  do {                                                                                                         \
    if (a.stamps && lane == 0) {                                                                               \
      for (int k = 0; k < 9; ++k) a.stamps[((int64_t)blockIdx.x * 4 + wave) * 11 + k] = stamp_sum[k];           \
      /* shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz */                                          \
      a.stamps[((int64_t)blockIdx.x * 4 + wave) * 11 + 9] = __builtin_amdgcn_s_memtime() - clk0;               \
      a.stamps[((int64_t)blockIdx.x * 4 + wave) * 11 + 10] = __builtin_amdgcn_s_memrealtime() - real0;         \
    }                                                                                                          \
  } while (0)
// device buffer of 4 * 11 * grid uint64, its address handed over by the diagnostic script
#define SMOS_STAMPS_HOST(a)                                                                            \
  do {                                                                                                 \
    const char* e_ = getenv("SMOS_CONV_STAMP_PTR");                                                    \
    (a).stamps = e_ ? reinterpret_cast<unsigned long long*>(strtoull(e_, nullptr, 0)) : nullptr;       \
  } while (0)
#define SMOS_STAMPS_HOST_NONE(a) (a).stamps = nullptr
#else
#define SMOS_STAMPS_ARG
#define SMOS_STAMPS_DECLARE() ((void)0)
#define SMOS_STAMP(k)
#define SMOS_STAMPS_BEGIN() ((void)0)
#define SMOS_STAMPS_END() ((void)0)
#define SMOS_STAMPS_HOST(a) ((void)0)
#define SMOS_STAMPS_HOST_NONE(a) ((void)0)
#endif

#if SMOS_CONV_SCHED == 0
// the stage of conv_igemm cut into four MFMA groups with the non-matrix work between them (see conv_igemm.hip for the names)
#define SMOS_STAGE(bc, bp, sc, sn, n0, n1, n2, n3) \
  do {                                             \
    SMOS_STAMP(8);                                 \
    mfma_group(af, bc[0], 0);                      \
    SMOS_FENCE();                                  \
    SMOS_STAMP(0);                                 \
    if (!(SMOS_CONV_ABLATE & 4)) {                 \
      park(sn, n0, n1, n2, n3);                    \
      load_a(n0, n1, n2, n3);                      \
      read_a(af, sc, 2);                           \
      read_a(af, sc, 3);                           \
    }                                              \
    SMOS_FENCE();                                  \
    SMOS_STAMP(1);                                 \
    mfma_group(af, bc[1], 1);                      \
    SMOS_FENCE();                                  \
    SMOS_STAMP(2);                                 \
    if (!(SMOS_CONV_ABLATE & 2)) load_b(bp);       \
    SMOS_FENCE();                                  \
    SMOS_STAMP(3);                                 \
    mfma_group(af, bc[2], 2);                      \
    SMOS_FENCE();                                  \
    SMOS_STAMP(4);                                 \
    advance_b();                                   \
    ring_barrier();                                \
    if (!(SMOS_CONV_ABLATE & 4)) {                 \
      read_a(af, sn, 0);                           \
      read_a(af, sn, 1);                           \
    }                                              \
    SMOS_FENCE();                                  \
    SMOS_STAMP(5);                                 \
    mfma_group(af, bc[3], 3);                      \
    SMOS_FENCE();                                  \
    SMOS_STAMP(6);                                 \
    if (--c_left == 0) {                           \
      epilogue();                                  \
      c_left = a.nstage;                           \
      ++c_it;                                      \
    }                                              \
    if (RES && c_left == 1) request_residual();    \
    if (pb_left == 0) next_tile_b();               \
    SMOS_STAMP(7);                                 \
  } while (0)
#endif
