// ipm_qp.h -- the barrier QP as the reference's stack solves it (`barrier_solver: cvxopt`, rg_scenario_params.qp_mode = 1):
// rps hands  min |u - uhat|^2  s.t.  -2 e_ij.u_i + 2 e_ij.u_j <= gamma h_ij^3  to cvxopt's interior-point `qp` at
// reltol = feastol = 1e-2, maxiters 50 (utilities/controller.py:13-16,23 -> rps barrier_certificates), and
// gets back an ITERATE that stops strictly inside the feasible set -- not the projection the default mode computes.  This file
// is that iteration (restated coneqp for the linear cone: default starting point, Mehrotra predictor-corrector, step 0.99 to the
// boundary, cvxopt's stopping rule) as the explicit sequence of IEEE binary64 operations of oracle/oracle_core.h
// barrier_qp_ipm_spec ("ipm_spec_v0"): same operations, same order, bit-identical results (tests/test_gpu_ipm.py).
//
// Binary64 inside a binary32 engine: the KKT matrix 2I + G' diag(z/s) G reaches condition numbers of 1e6 .. 1e8 and cvxopt's
// stopping rule holds the dual residual against an absolute 1e-2 while the multipliers are ~1e4; a binary32 transcription misses
// the stop and runs into NaN (measured on the CPU twin).  gfx950 issues v_fma_f64 at the v_fma_f32 rate: the cost is registers.
//
// Mapping: the whole QP of one env runs IN ONE LANE (2N unknowns, N(N-1)/2 rows, everything in registers, every loop unrolled).
// The lane-group kernel gathers an env's xi / uhat through LDS and lets every lane of the group run the same iteration (the
// lanes of a group are otherwise idle during the QP, and a distributed factorisation of a 10 x 10 .. 16 x 16 matrix costs more
// exchanges than it saves multiplications); the thread-per-env kernel calls it per lane.  One out-of-line body per agent count
// and translation unit (noinline): the iteration is ~2 k (N = 5) .. 8 k (N = 8) instructions and would otherwise be copied into
// every kernel instantiation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/robogym.h"

namespace rg {
namespace ipm {

constexpr int MAX_N = 8;

struct Consts {
    double abstol, reltol, feas2, r2, gain, ugain;
    int maxiters, has_unsafe;
};
// binary32 parameters widened exactly; products formed in binary64 (the CPU twin does the same on the same binary32 values)
__host__ __device__ inline Consts make_consts(const rg_scenario_params &p) {
    Consts k;
    k.abstol = static_cast<double>(p.ipm_abstol);
    k.reltol = static_cast<double>(p.ipm_reltol);
    k.feas2 = static_cast<double>(p.ipm_feastol) * static_cast<double>(p.ipm_feastol);
    k.r2 = static_cast<double>(p.safety_radius) * static_cast<double>(p.safety_radius);
    k.gain = static_cast<double>(p.barrier_gain);
    k.ugain = static_cast<double>(p.unsafe_barrier_gain);
    k.maxiters = p.ipm_maxiters;
    k.has_unsafe = p.barrier_has_unsafe_gain;
    return k;
}

// packed lower triangle
__host__ __device__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }

// K <- 2I + sum_c w_c a_c a_c'  (a_c = -2 e_c at robot i, +2 e_c at robot j; w4 = 4 w): diagonal blocks sum over the partners in
// row order, an off-diagonal block is -w4 e e' of its one pair
template <int N>
__device__ __forceinline__ void assemble(const double (&ex)[N * (N - 1) / 2], const double (&ey)[N * (N - 1) / 2],
                                         const double (&w4)[N * (N - 1) / 2], double (&K)[N * (2 * N + 1)]) {
    constexpr int n = 2 * N;
#pragma unroll
    for (int r = 0; r < n; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) K[tri(r, c)] = r == c ? 2.0 : 0.0;
    int c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i)
#pragma unroll
        for (int j = i + 1; j < N; ++j, ++c) {
            const double a = w4[c] * ex[c], b = w4[c] * ey[c];
            const double wxx = a * ex[c], wxy = a * ey[c], wyy = b * ey[c];
            K[tri(2 * i, 2 * i)] = K[tri(2 * i, 2 * i)] + wxx;
            K[tri(2 * i + 1, 2 * i)] = K[tri(2 * i + 1, 2 * i)] + wxy;
            K[tri(2 * i + 1, 2 * i + 1)] = K[tri(2 * i + 1, 2 * i + 1)] + wyy;
            K[tri(2 * j, 2 * j)] = K[tri(2 * j, 2 * j)] + wxx;
            K[tri(2 * j + 1, 2 * j)] = K[tri(2 * j + 1, 2 * j)] + wxy;
            K[tri(2 * j + 1, 2 * j + 1)] = K[tri(2 * j + 1, 2 * j + 1)] + wyy;
            K[tri(2 * j, 2 * i)] = -wxx;
            K[tri(2 * j, 2 * i + 1)] = -wxy;
            K[tri(2 * j + 1, 2 * i)] = -wxy;
            K[tri(2 * j + 1, 2 * i + 1)] = -wyy;
        }
}
// in place: K <- unit lower L (below the diagonal) and D (on it); rd <- 1 / D
template <int N>
__device__ __forceinline__ void ldl(double (&K)[N * (2 * N + 1)], double (&rd)[2 * N]) {
    constexpr int n = 2 * N;
    double v[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double d = K[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            v[k] = K[tri(j, k)] * K[tri(k, k)];
            d = __builtin_fma(-K[tri(j, k)], v[k], d);
        }
        K[tri(j, j)] = d;
        rd[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double t = K[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t = __builtin_fma(-K[tri(i, k)], v[k], t);
            K[tri(i, j)] = t * rd[j];
        }
    }
}
template <int N>
__device__ __forceinline__ void solve(const double (&K)[N * (2 * N + 1)], const double (&rd)[2 * N], double (&b)[2 * N]) {
    constexpr int n = 2 * N;
#pragma unroll
    for (int i = 1; i < n; ++i) {
        double t = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t = __builtin_fma(-K[tri(i, k)], b[k], t);
        b[i] = t;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) b[i] = b[i] * rd[i];
#pragma unroll
    for (int i = n - 2; i >= 0; --i) {
        double t = b[i];
#pragma unroll
        for (int k = i + 1; k < n; ++k) t = __builtin_fma(-K[tri(k, i)], b[k], t);
        b[i] = t;
    }
}

// io: N records (xi_x, xi_y, uhat_x, uhat_y) of the env, uhat already thresholded to the magnitude limit; the iterate replaces
// uhat.  Returns cvxopt's `iterations`.  `io` is a generic pointer (LDS in both kernels).
// Register budget: callers are one-wave workgroups; the body is compiled for one wave per SIMD (512 registers: 256 + 256
// accumulation registers as spill space) -- without the attributes a non-kernel function is held to the default 128.
template <int N>
__device__ __attribute__((noinline)) int solve_qp(const Consts k, float4 *io) {
    static_assert(N >= 2 && N <= MAX_N, "agent count");
    constexpr int n = 2 * N, m = N * (N - 1) / 2;
    double ex[m], ey[m], h[m], s[m], z[m], w4[m], rz[m], rs[m], ds[m], dz[m], dsdza[m];
    double q[n], x[n], rx[n], dx[n], rd[n], K[N * (2 * N + 1)];
    double nh = 0.0, nq = 0.0;
    {
        double xix[N], xiy[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const float4 r = io[a];
            xix[a] = static_cast<double>(r.x);
            xiy[a] = static_cast<double>(r.y);
            q[2 * a] = -2.0 * static_cast<double>(r.z);
            q[2 * a + 1] = -2.0 * static_cast<double>(r.w);
            nq = __builtin_fma(q[2 * a], q[2 * a], nq);
            nq = __builtin_fma(q[2 * a + 1], q[2 * a + 1], nq);
        }
        int c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; ++i)
#pragma unroll
            for (int j = i + 1; j < N; ++j, ++c) {
                ex[c] = xix[i] - xix[j];
                ey[c] = xiy[i] - xiy[j];
                const double hh = __builtin_fma(ex[c], ex[c], ey[c] * ey[c]) - k.r2;
                const double gain = (hh >= 0.0 || !k.has_unsafe) ? k.gain : k.ugain;
                h[c] = gain * ((hh * hh) * hh);
                nh = __builtin_fma(h[c], h[c], nh);
                w4[c] = 4.0;
            }
    }
    const double resx0sq = k.feas2 * (nq > 1.0 ? nq : 1.0), resz0sq = k.feas2 * (nh > 1.0 ? nh : 1.0);
    // default starting point: (2I + G'G) x = -q + G'h;  z = G x - h;  s = -z;  both shifted into the cone if they are not inside
    assemble<N>(ex, ey, w4, K);
    ldl<N>(K, rd);
#pragma unroll
    for (int kk = 0; kk < n; ++kk) x[kk] = -q[kk];
    {
        int c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; ++i)
#pragma unroll
            for (int j = i + 1; j < N; ++j, ++c) {
                const double t = 2.0 * h[c];
                x[2 * i] = __builtin_fma(-t, ex[c], x[2 * i]);
                x[2 * i + 1] = __builtin_fma(-t, ey[c], x[2 * i + 1]);
                x[2 * j] = __builtin_fma(t, ex[c], x[2 * j]);
                x[2 * j + 1] = __builtin_fma(t, ey[c], x[2 * j + 1]);
            }
    }
    solve<N>(K, rd, x);
    double gap;
    {
        double ns = 0.0, tz = -1e300, ts = -1e300;
        int c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; ++i)
#pragma unroll
            for (int j = i + 1; j < N; ++j, ++c) {
                const double gx_ = 2.0 * __builtin_fma(ex[c], x[2 * j] - x[2 * i], ey[c] * (x[2 * j + 1] - x[2 * i + 1]));
                z[c] = gx_ - h[c];
                s[c] = -z[c];
                ns = __builtin_fma(z[c], z[c], ns);
                ts = z[c] > ts ? z[c] : ts;  // max(-s)
                tz = s[c] > tz ? s[c] : tz;  // max(-z)
            }
        // t >= -1e-8 max(|s|, 1), without the root: t >= 0, or t^2 <= 1e-16 max(s.s, 1)
        const double lim2 = 1e-16 * (ns > 1.0 ? ns : 1.0);
        const bool shift_s = ts >= 0.0 || ts * ts <= lim2, shift_z = tz >= 0.0 || tz * tz <= lim2;
        const double as = 1.0 + ts, az = 1.0 + tz;
        gap = 0.0;
#pragma unroll
        for (c = 0; c < m; ++c) {
            s[c] = shift_s ? s[c] + as : s[c];
            z[c] = shift_z ? z[c] + az : z[c];
            gap = __builtin_fma(s[c], z[c], gap);
        }
    }
    int iters = 0;
    for (;; ++iters) {
        // residuals: rx = q + 2x + G'z, rz = s - h + G x; costs
        double f0 = 0.0, nrx = 0.0, nrz = 0.0, zrz = 0.0;
#pragma unroll
        for (int kk = 0; kk < n; ++kk) {
            rx[kk] = __builtin_fma(2.0, x[kk], q[kk]);
            f0 = __builtin_fma(x[kk], rx[kk] + q[kk], f0);
        }
        f0 = 0.5 * f0;
        {
            int c = 0;
#pragma unroll
            for (int i = 0; i < N - 1; ++i)
#pragma unroll
                for (int j = i + 1; j < N; ++j, ++c) {
                    const double t = 2.0 * z[c];
                    rx[2 * i] = __builtin_fma(-t, ex[c], rx[2 * i]);
                    rx[2 * i + 1] = __builtin_fma(-t, ey[c], rx[2 * i + 1]);
                    rx[2 * j] = __builtin_fma(t, ex[c], rx[2 * j]);
                    rx[2 * j + 1] = __builtin_fma(t, ey[c], rx[2 * j + 1]);
                    const double gx_ = 2.0 * __builtin_fma(ex[c], x[2 * j] - x[2 * i], ey[c] * (x[2 * j + 1] - x[2 * i + 1]));
                    rz[c] = (s[c] - h[c]) + gx_;
                    nrz = __builtin_fma(rz[c], rz[c], nrz);
                    zrz = __builtin_fma(z[c], rz[c], zrz);
                }
        }
#pragma unroll
        for (int kk = 0; kk < n; ++kk) nrx = __builtin_fma(rx[kk], rx[kk], nrx);
        const double pcost = f0, dcost = (f0 + zrz) - gap;
        const bool rel_ok = pcost < 0.0 ? gap <= k.reltol * -pcost : dcost > 0.0 ? gap <= k.reltol * dcost : false;
        if ((nrz <= resz0sq && nrx <= resx0sq && (gap <= k.abstol || rel_ok)) || iters == k.maxiters) break;
        // scaling and the KKT matrix of this iteration
#pragma unroll
        for (int c = 0; c < m; ++c) {
            rs[c] = 1.0 / s[c];
            w4[c] = 4.0 * (z[c] * rs[c]);
        }
        assemble<N>(ex, ey, w4, K);
        ldl<N>(K, rd);
        const double mu = gap / static_cast<double>(m);
        double sigmamu = 0.0, step = 1.0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int kk = 0; kk < n; ++kk) dx[kk] = -rx[kk];
            {
                int c = 0;
#pragma unroll
                for (int i = 0; i < N - 1; ++i)
#pragma unroll
                    for (int j = i + 1; j < N; ++j, ++c) {
                        const double rc = pass ? __builtin_fma(-s[c], z[c], sigmamu) - dsdza[c] : -(s[c] * z[c]);
                        const double t = 2.0 * (__builtin_fma(z[c], rz[c], rc) * rs[c]);
                        dx[2 * i] = __builtin_fma(t, ex[c], dx[2 * i]);
                        dx[2 * i + 1] = __builtin_fma(t, ey[c], dx[2 * i + 1]);
                        dx[2 * j] = __builtin_fma(-t, ex[c], dx[2 * j]);
                        dx[2 * j + 1] = __builtin_fma(-t, ey[c], dx[2 * j + 1]);
                    }
            }
            solve<N>(K, rd, dx);
            double dsdz = 0.0, tn = 0.0, td = 1.0;
            {
                int c = 0;
#pragma unroll
                for (int i = 0; i < N - 1; ++i)
#pragma unroll
                    for (int j = i + 1; j < N; ++j, ++c) {
                        const double gdx = 2.0 * __builtin_fma(ex[c], dx[2 * j] - dx[2 * i], ey[c] * (dx[2 * j + 1] - dx[2 * i + 1]));
                        const double rc = pass ? __builtin_fma(-s[c], z[c], sigmamu) - dsdza[c] : -(s[c] * z[c]);
                        ds[c] = -rz[c] - gdx;
                        dz[c] = __builtin_fma(-z[c], ds[c], rc) * rs[c];
                        dsdz = __builtin_fma(ds[c], dz[c], dsdz);
                        if (-ds[c] * td > tn * s[c]) {
                            tn = -ds[c];
                            td = s[c];
                        }
                        if (-dz[c] * td > tn * z[c]) {
                            tn = -dz[c];
                            td = z[c];
                        }
                    }
            }
            if (pass == 0) {
                step = tn > td ? td / tn : 1.0;
                double sg = __builtin_fma(dsdz / gap, step * step, 1.0 - step);
                sg = sg < 0.0 ? 0.0 : sg > 1.0 ? 1.0 : sg;
                sigmamu = ((sg * sg) * sg) * mu;
#pragma unroll
                for (int c = 0; c < m; ++c) dsdza[c] = ds[c] * dz[c];
            } else {
                step = 0.99 * td < tn ? (0.99 * td) / tn : 1.0;
            }
        }
#pragma unroll
        for (int kk = 0; kk < n; ++kk) x[kk] = __builtin_fma(step, dx[kk], x[kk]);
        gap = 0.0;
#pragma unroll
        for (int c = 0; c < m; ++c) {
            s[c] = __builtin_fma(step, ds[c], s[c]);
            z[c] = __builtin_fma(step, dz[c], z[c]);
            gap = __builtin_fma(s[c], z[c], gap);
        }
    }
#pragma unroll
    for (int a = 0; a < N; ++a) {
        float4 r = io[a];
        r.z = static_cast<float>(x[2 * a]);
        r.w = static_cast<float>(x[2 * a + 1]);
        io[a] = r;
    }
    return iters;
}

// runtime agent count -> the out-of-line body (N = 1: no rows, the unconstrained minimiser is the thresholded input itself)
__device__ __forceinline__ int solve_qp_n(int N, const Consts &k, float4 *io) {
    switch (N) {
        case 2: return solve_qp<2>(k, io);
        case 3: return solve_qp<3>(k, io);
        case 4: return solve_qp<4>(k, io);
        case 5: return solve_qp<5>(k, io);
        case 6: return solve_qp<6>(k, io);
        case 7: return solve_qp<7>(k, io);
        case 8: return solve_qp<8>(k, io);
        default: return 0;
    }
}

}  // namespace ipm
}  // namespace rg
