// probes/diag.h -- every diagnostic hook of the two step kernels, in one place.
//
// The shipped library is built with none of the switches below: every hook then expands to nothing (or to the plain statement
// it wraps) and step_group.h / step_tpe.h read as the product.  Diagnostic builds (tools/README.md) define:
//   RG_STAMPS                wave-cycle stamps of the step's phases, written over io.qp_sweeps of the wave's first 8 envs
//     + RG_STAMPS_EPI        slots 0..2 mark points inside the PredatorCapturePrey epilogue instead
//     + RG_STAMPS_CLOCK      slot 5 = the wave's life on the constant 100 MHz clock (slot 6 / slot 5 = shader clock / 100 MHz)
//     + RG_STAMPS_CTRL       ticks per controller part (set-up / sweeps / tail / heading sin-cos), summed over the step
//     + RG_STAMPS_CHUNK      ticks inside the dense pre-test and the exact replay, and how often each runs
//   RG_TPE_DIAG              thread-per-env kernel: replayed-chunk mask and per-controller sweep counts in io.qp_sweeps
//   RG_TPE_GUARD             thread-per-env kernel: stores through the LDS staging block are bounds-checked
//   RG_HOST_SIM              the kernels compiled for the host (tests/sanitize/): 64 threads = 64 lanes
// and the tuning constants RG_CHUNK, RG_MAX_WAVES, RG_TPE_W4 / W5 / W78, RG_TPE_NO_W3 (A/B builds).
// One-off probes of rounds 2-4 (shadow copies, RG_PROBE_*, RG_FIXED_U, RG_DENSE_PRETEST, RG_NT_STORES, RG_NO_XCD_REMAP,
// RG_TPE_NO_DMAX_GUARD) were removed in round 5; what they measured is in NOTEBOOK.md and profiles/, the code in git history.
#pragma once

// ---- tuning constants (defaults = the shipped build)
#ifndef RG_CHUNK
#define RG_CHUNK 5      // lane-group kernel: sub-steps validated together
#endif
#ifndef RG_MAX_WAVES
#define RG_MAX_WAVES 1024   // lane-group kernel: partly filled waves up to one wave per SIMD
#endif
#ifndef RG_TPE_W4
#define RG_TPE_W4 3   // waves per SIMD the N <= 4 instantiations are compiled for
#endif
#ifndef RG_TPE_W5
#define RG_TPE_W5 0   // N = 5 (0 = the compiler's own allocation: two)
#endif
#ifndef RG_TPE_W78
#define RG_TPE_W78 1  // N = 7, 8
#endif
#ifndef RG_TPE_NO_W3
#define RG_TPE_WAVES_ATTR(n) __attribute__((amdgpu_waves_per_eu(n)))
#else
#define RG_TPE_WAVES_ATTR(n)
#endif

namespace rg {
#ifdef RG_STAMPS
constexpr bool kStampsBuild = true;
#else
constexpr bool kStampsBuild = false;
#endif
#ifdef RG_HOST_SIM
constexpr bool kHostSim = true;
#else
constexpr bool kHostSim = false;
#endif
#ifdef RG_TPE_GUARD
constexpr bool kTpeGuard = true;
#else
constexpr bool kTpeGuard = false;
#endif
}  // namespace rg

// ---- phase stamps (both kernels)
#ifdef RG_STAMPS
#define RG_STAMPS_BEGIN()                                                    \
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();         \
    int stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                \
    RG_STAMPS_CLOCK_BEGIN()
#define RG_STAMP_ALWAYS_5() stamps[5] = static_cast<int>(__builtin_amdgcn_s_memtime() - t_start)   /* thread-per-env kernel: end of the step */
#define RG_PIN1(a) asm volatile("" ::"v"(a))
#define RG_PIN2(a, b) asm volatile("" ::"v"(a), "v"(b))
#define RG_PIN3(a, b, c) asm volatile("" ::"v"(a), "v"(b), "v"(c))
#define RG_PIN4(a, b, c, d) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d))
#define RG_PIN5(a, b, c, d, e) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e))
#ifdef RG_STAMPS_EPI
#define RG_STAMP(i) \
    if ((i) > 2) stamps[i] = static_cast<int>(__builtin_amdgcn_s_memtime() - t_start)
#define RG_STAMP_E(i) stamps[i] = static_cast<int>(__builtin_amdgcn_s_memtime() - t_start)
#else
#define RG_STAMP(i) stamps[i] = static_cast<int>(__builtin_amdgcn_s_memtime() - t_start)
#define RG_STAMP_E(i)
#endif
#ifdef RG_STAMPS_CLOCK
#define RG_STAMPS_CLOCK_BEGIN() const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#define RG_STAMPS_CLOCK_END() stamps[5] = static_cast<int>(__builtin_amdgcn_s_memrealtime() - rt_start);
#else
#define RG_STAMPS_CLOCK_BEGIN()
#define RG_STAMPS_CLOCK_END()
#endif
// the stamps go out over io.qp_sweeps of the wave's first 8 envs (e0 = the wave's first env)
#define RG_STAMPS_WRITE(lane_, qp_, e0_, E_, max_sweeps_)                                     \
    RG_STAMPS_CLOCK_END()                                                                     \
    if ((lane_) == 0 && (qp_)) {                                                              \
        stamps[7] = (max_sweeps_);                                                            \
        for (int i_ = 0; i_ < 8; ++i_)                                                        \
            if ((e0_) + i_ < (E_)) (qp_)[(e0_) + i_] = stamps[i_];                            \
    }
#else
#define RG_STAMPS_BEGIN()
#define RG_STAMP_ALWAYS_5()
#define RG_PIN1(a)
#define RG_PIN2(a, b)
#define RG_PIN3(a, b, c)
#define RG_PIN4(a, b, c, d)
#define RG_PIN5(a, b, c, d, e)
#define RG_STAMP(i)
#define RG_STAMP_E(i)
#define RG_STAMPS_WRITE(lane_, qp_, e0_, E_, max_sweeps_)
#endif

// ---- lane-group kernel: ticks per controller part (RG_STAMPS_CTRL) and per pre-test fall-back (RG_STAMPS_CHUNK)
#if defined(RG_STAMPS) && defined(RG_STAMPS_CTRL)
#define RG_CTRL_TICKS_PARAM , int *ctrl_ticks = nullptr
#define RG_CTRL_TICKS_ARG , ctrl_ticks
#define RG_CTRL_LOCALS() int ctrl_ticks[5] = {0, 0, 0, 0, 0};  // set-up, sweeps, tail, wave-level sweep count, heading sin/cos
#define RG_CTRL_BEGIN(x, y, c, s)                      \
    asm volatile("" ::"v"(x), "v"(y), "v"(c), "v"(s)); \
    unsigned long long ctrl_t = __builtin_amdgcn_s_memtime();
#define RG_CTRL_TICK(i)                                                     \
    {                                                                       \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        ctrl_ticks[i] += static_cast<int>(t_ - ctrl_t);                     \
        ctrl_t = t_;                                                        \
    }
#define RG_CTRL_PIN_ROUNDS(GW_, fx, bp, emax) \
    _Pragma("unroll") for (int r_ = 0; r_ < (GW_) - 1; ++r_) asm volatile("" ::"v"(fx[r_]), "v"(bp[r_]), "v"(emax[r_]));
#define RG_CTRL_PIN2(a, b) asm volatile("" ::"v"(a), "v"(b))
#define RG_CTRL_COUNT_SWEEPS(n) ctrl_ticks[3] += (n)
// the heading's sin / cos, timed
#define RG_HEADING_SINCOS(th, s, c)                                                   \
    asm volatile("" ::"v"(th));                                                       \
    const unsigned long long t_sc = __builtin_amdgcn_s_memtime();                     \
    sincos_spec(th, s, c);                                                            \
    asm volatile("" ::"v"(s), "v"(c));                                                \
    ctrl_ticks[4] += static_cast<int>(__builtin_amdgcn_s_memtime() - t_sc)
#define RG_CTRL_REPORT()                                                                                        \
    stamps[0] = ctrl_ticks[0];                                                                                  \
    stamps[1] = ctrl_ticks[1];                                                                                  \
    stamps[2] = ctrl_ticks[2] * 100 + ctrl_ticks[3]; /* tail ticks x 100 + sweeps the wave executed */          \
    stamps[3] = ctrl_ticks[4];
#else
#define RG_CTRL_TICKS_PARAM
#define RG_CTRL_TICKS_ARG
#define RG_CTRL_LOCALS()
#define RG_CTRL_BEGIN(x, y, c, s)
#define RG_CTRL_TICK(i)
#define RG_CTRL_PIN_ROUNDS(GW_, fx, bp, emax)
#define RG_CTRL_PIN2(a, b)
#define RG_CTRL_COUNT_SWEEPS(n)
#define RG_HEADING_SINCOS(th, s, c) sincos_spec(th, s, c)
#define RG_CTRL_REPORT()
#endif

#if defined(RG_STAMPS) && defined(RG_STAMPS_CHUNK)
#define RG_CHUNK_LOCALS() int chunk_ticks[4] = {0, 0, 0, 0};  // ticks in the dense pre-test, ticks in the replay, dense chunks, replayed chunks
#define RG_CHUNK_DENSE_BEGIN() const unsigned long long t_dense0 = __builtin_amdgcn_s_memtime();
#define RG_CHUNK_DENSE_END(dmin)                                          \
    asm volatile("" ::"v"(dmin));                                         \
    const unsigned long long t_dense1 = __builtin_amdgcn_s_memtime();     \
    chunk_ticks[0] += static_cast<int>(t_dense1 - t_dense0);              \
    chunk_ticks[2] += 1;
#define RG_CHUNK_REPLAY_END(rx, ry)                                                    \
    asm volatile("" ::"v"(rx), "v"(ry));                                               \
    chunk_ticks[1] += static_cast<int>(__builtin_amdgcn_s_memtime() - t_dense1);       \
    chunk_ticks[3] += 1;
#define RG_CHUNK_REPORT()           \
    stamps[0] = chunk_ticks[0];     \
    stamps[1] = chunk_ticks[1];     \
    stamps[2] = chunk_ticks[2] * 1000 + chunk_ticks[3];
#else
#define RG_CHUNK_LOCALS()
#define RG_CHUNK_DENSE_BEGIN()
#define RG_CHUNK_DENSE_END(dmin)
#define RG_CHUNK_REPLAY_END(rx, ry)
#define RG_CHUNK_REPORT()
#endif

// ---- thread-per-env kernel
#if defined(RG_STAMPS) && !defined(RG_STAMPS_EPI)
#define RG_TSTAMP_MAIN(i) RG_STAMP(i)
#define RG_TSTAMP_EPI(i)
#define RG_TSTAMP_PERIOD(it0) \
    if ((it0) == 0) RG_STAMP(2)   /* first period */
#elif defined(RG_STAMPS)
#define RG_TSTAMP_MAIN(i)
#define RG_TSTAMP_EPI(i) RG_STAMP_E(i)
#define RG_TSTAMP_PERIOD(it0) \
    if ((it0) != 0) RG_STAMP_E(2) /* last period (the epilogue's loads are issued after it) */
#else
#define RG_TSTAMP_MAIN(i)
#define RG_TSTAMP_EPI(i)
#define RG_TSTAMP_PERIOD(it0)
#endif
#ifdef RG_STAMPS
#define RG_PIN_ARR1(N_, a) _Pragma("unroll") for (int i_ = 0; i_ < (N_); ++i_) asm volatile("" ::"v"(a[i_]))
#define RG_PIN_ARR2(N_, a, b) _Pragma("unroll") for (int i_ = 0; i_ < (N_); ++i_) asm volatile("" ::"v"(a[i_]), "v"(b[i_]))
#define RG_PIN_ARR3(N_, a, b, c) _Pragma("unroll") for (int i_ = 0; i_ < (N_); ++i_) asm volatile("" ::"v"(a[i_]), "v"(b[i_]), "v"(c[i_]))
#define RG_PIN_ARR5(N_, a, b, c, d, e) \
    _Pragma("unroll") for (int i_ = 0; i_ < (N_); ++i_) asm volatile("" ::"v"(a[i_]), "v"(b[i_]), "v"(c[i_]), "v"(d[i_]), "v"(e[i_]))
#else
#define RG_PIN_ARR1(N_, a)
#define RG_PIN_ARR2(N_, a, b)
#define RG_PIN_ARR3(N_, a, b, c)
#define RG_PIN_ARR5(N_, a, b, c, d, e)
#endif
#ifdef RG_TPE_DIAG
#define RG_TPE_DIAG_LOCALS() int diag = 0;
#define RG_TPE_DIAG_SWEEPS(sw, it0) diag |= (sw) << ((it0) == 0 ? 8 : 0)
#define RG_TPE_DIAG_REPLAY(it0, j0, CH_) diag |= 1 << (16 + ((it0) ? 3 : 0) + (j0) / (CH_))
#define RG_TPE_DIAG_REPORT(max_sweeps) max_sweeps = diag   // replayed-chunk mask << 16 | sweeps of QP 1 << 8 | sweeps of QP 2
#else
#define RG_TPE_DIAG_LOCALS()
#define RG_TPE_DIAG_SWEEPS(sw, it0)
#define RG_TPE_DIAG_REPLAY(it0, j0, CH_)
#define RG_TPE_DIAG_REPORT(max_sweeps)
#endif
// RG_TPE_GUARD: a store outside its array is dropped and flagged in done_count[0] instead of faulting (tests/guard_probe.py)
#ifdef RG_TPE_GUARD
#define RG_GUARDED(dst, lo, hi, code, stmt) \
    if ((dst) < (lo) || (dst) >= (hi)) atomicOr(sg.flag, (code)); else { stmt; }
#else
#define RG_GUARDED(dst, lo, hi, code, stmt) stmt
#endif

// ---- interior-point iteration (csrc/ipm_qp.h): -DRG_IPM_STAMPS accumulates the wave's ticks per phase of an iteration in
// `ipm_ticks[8]` (a __device__ array the diagnostic driver reads): 0 row phase, 1 residuals + stopping rule, 2 assembly,
// 3 factorisation, 4 right-hand side + solve (both passes), 5 ds / dz rows (both), 6 step to the boundary (both), 7 update
#ifdef RG_IPM_STAMPS
namespace rg { namespace ipm { __device__ unsigned long long ipm_ticks[8]; } }
#define RG_IPM_T0() unsigned long long ipm_t_ = __builtin_amdgcn_s_memtime();
#define RG_IPM_TICK(i)                                                                              \
    {                                                                                               \
        const unsigned long long t2_ = __builtin_amdgcn_s_memtime();                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) rg::ipm::ipm_ticks[i] += t2_ - ipm_t_;             \
        ipm_t_ = t2_;                                                                               \
    }
#else
#define RG_IPM_T0()
#define RG_IPM_TICK(i)
#endif
