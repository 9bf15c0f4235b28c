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
// Mapping.  The 2N unknowns (x, the residuals, the KKT matrix K and its factor) live in REGISTERS, replicated in every lane that
// works on the env; the N(N-1)/2 rows (e, h, s, z, rz, 1/s, ds, dz, ...) live in the env's LDS workspace.  A group of GS lanes
// shares one env (GS = the lane-group kernel's group width; 1 = one lane per env):
//   * row phases (G x, the scaling, the right-hand side's row factors, ds / dz, the update of s and z) are DISTRIBUTED: lane `sub`
//     of the group takes rows sub, sub + GS, ... -- each row is computed once, by one lane, and published in LDS;
//   * everything that accumulates (G' z into the residual, the sums of the stopping rule, K's blocks, the step to the boundary)
//     and the factorisation / solves run REPLICATED in every lane of the group, reading the published rows in row order -- so every
//     sum is formed in exactly the order of the CPU twin whatever GS is, and all lanes of a group hold identical values and take
//     identical branches.
// LDS operations of a wavefront execute in program order, so a lane reads what another lane of its group published without a
// barrier; wavefront-scope fences keep the compiler from reordering across a phase boundary.  One out-of-line body per (N, GS) and
// translation unit (noinline): an iteration is ~1.5 k (N = 5) .. 4 k (N = 8) instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_common.h"
#include "probes/diag.h"

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

// 1 / v for v > 0 (oracle_core.h ipm_rcp): exponent-field seed + five Newton steps in fma -- deterministic (the same bits as the
// CPU twin, which v_rcp_f64 cannot promise), within two ulp, 12 plain VALU instructions that interleave with their neighbours,
// against ~30 dependent ones through VCC for a correctly rounded binary64 division
__device__ __forceinline__ double rcp_spec(double v) {
    const unsigned long long b = 0x7FDE6238DA3C2118ull - __builtin_bit_cast(unsigned long long, v);
    double r = __builtin_bit_cast(double, b);
#pragma unroll
    for (int t = 0; t < 5; ++t) r = __builtin_fma(r, __builtin_fma(-v, r, 1.0), r);
    return r;
}

// packed lower triangle
__host__ __device__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }
// rows in rps' order (i < j, i outer): row c <-> (i, j)
__host__ __device__ constexpr int row_i(int N, int c) {
    int i = 0;
    while (c >= N - 1 - i) {
        c -= N - 1 - i;
        ++i;
    }
    return i;
}
__host__ __device__ constexpr int row_j(int N, int c) {
    int i = 0;
    while (c >= N - 1 - i) {
        c -= N - 1 - i;
        ++i;
    }
    return i + 1 + c;
}
// slot q of a group of GS lanes: lane `sub` works on row q * GS + sub; its (i, j) as 3 + 3 bits per lane in one 64-bit constant
template <int N, int GS>
__host__ __device__ constexpr unsigned long long slot_table(int q) {
    unsigned long long t = 0;
    constexpr int m = N * (N - 1) / 2;
    for (int s = 0; s < GS; ++s) {
        int c = q * GS + s;
        c = c < m ? c : m - 1;
        t |= static_cast<unsigned long long>(row_i(N, c) | (row_j(N, c) << 3)) << (6 * s);
    }
    return t;
}

typedef __attribute__((address_space(3))) double lds_f64;
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f64x2 lds_f64x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

// The env's LDS workspace (doubles): per row (e_x, e_y) | h | (s, z) | (rz, 1/s) | (ds, dz) | dsa dza | t | 4w, then the
// published copies of x and dx.  16-byte fields are read and written whole.
template <int N>
struct Ws {
    static constexpr int m = N * (N - 1) / 2, n = 2 * N;
    static constexpr int E = 0, SZ = 2 * m, RR = 4 * m, DD = 6 * m, H = 8 * m, A = 9 * m, T = 10 * m, W = 11 * m, X = 12 * m, DX = 12 * m + n;
    static constexpr int SIZE = 12 * m + 2 * n;   // doubles (even: the next env's block stays 16-byte aligned)
};
constexpr int ws_doubles(int N) { return 12 * (N * (N - 1) / 2) + 4 * N; }

// Where an env's rows (and the published copies of x / dx) live.  LdsRows: the env's LDS workspace, shared by the GS lanes of a
// group, any row addressable at run time.  RegRows: this lane's registers -- for GS = 1 (every lane computes every row; the row
// index is a compile-time constant after unrolling): cheaper than LDS while the rows fit, i.e. for N <= 4 (and N = 5 per lane).
template <int N>
struct LdsRows {
    typedef Ws<N> L;
    lds_f64 *ws;
    __device__ __forceinline__ f64x2 e(int c) const { return *reinterpret_cast<const lds_f64x2 *>(ws + L::E + 2 * c); }
    __device__ __forceinline__ void set_e(int c, f64x2 v) { *reinterpret_cast<lds_f64x2 *>(ws + L::E + 2 * c) = v; }
    __device__ __forceinline__ f64x2 sz(int c) const { return *reinterpret_cast<const lds_f64x2 *>(ws + L::SZ + 2 * c); }
    __device__ __forceinline__ double s(int c) const { return ws[L::SZ + 2 * c]; }
    __device__ __forceinline__ void set_sz(int c, f64x2 v) { *reinterpret_cast<lds_f64x2 *>(ws + L::SZ + 2 * c) = v; }
    __device__ __forceinline__ f64x2 rr(int c) const { return *reinterpret_cast<const lds_f64x2 *>(ws + L::RR + 2 * c); }
    __device__ __forceinline__ double rz(int c) const { return ws[L::RR + 2 * c]; }
    __device__ __forceinline__ void set_rr(int c, f64x2 v) { *reinterpret_cast<lds_f64x2 *>(ws + L::RR + 2 * c) = v; }
    __device__ __forceinline__ f64x2 dd(int c) const { return *reinterpret_cast<const lds_f64x2 *>(ws + L::DD + 2 * c); }
    __device__ __forceinline__ void set_dd(int c, f64x2 v) { *reinterpret_cast<lds_f64x2 *>(ws + L::DD + 2 * c) = v; }
    __device__ __forceinline__ double h(int c) const { return ws[L::H + c]; }
    __device__ __forceinline__ void set_h(int c, double v) { ws[L::H + c] = v; }
    __device__ __forceinline__ double a(int c) const { return ws[L::A + c]; }
    __device__ __forceinline__ void set_a(int c, double v) { ws[L::A + c] = v; }
    __device__ __forceinline__ double t(int c) const { return ws[L::T + c]; }
    __device__ __forceinline__ void set_t(int c, double v) { ws[L::T + c] = v; }
    __device__ __forceinline__ double w(int c) const { return ws[L::W + c]; }
    __device__ __forceinline__ void set_w(int c, double v) { ws[L::W + c] = v; }
    __device__ __forceinline__ f64x2 x(int a_) const { return *reinterpret_cast<const lds_f64x2 *>(ws + L::X + 2 * a_); }
    __device__ __forceinline__ void set_x(int a_, f64x2 v) { *reinterpret_cast<lds_f64x2 *>(ws + L::X + 2 * a_) = v; }
    __device__ __forceinline__ f64x2 dx(int a_) const { return *reinterpret_cast<const lds_f64x2 *>(ws + L::DX + 2 * a_); }
    __device__ __forceinline__ void set_dx(int a_, f64x2 v) { *reinterpret_cast<lds_f64x2 *>(ws + L::DX + 2 * a_) = v; }
};
template <int N>
struct RegRows {
    static constexpr int m = N * (N - 1) / 2;
    f64x2 e_[m], sz_[m], rr_[m], dd_[m], x_[N], dx_[N];
    double h_[m], a_[m], t_[m], w_[m];
    __device__ __forceinline__ f64x2 e(int c) const { return e_[c]; }
    __device__ __forceinline__ void set_e(int c, f64x2 v) { e_[c] = v; }
    __device__ __forceinline__ f64x2 sz(int c) const { return sz_[c]; }
    __device__ __forceinline__ double s(int c) const { return sz_[c].x; }
    __device__ __forceinline__ void set_sz(int c, f64x2 v) { sz_[c] = v; }
    __device__ __forceinline__ f64x2 rr(int c) const { return rr_[c]; }
    __device__ __forceinline__ double rz(int c) const { return rr_[c].x; }
    __device__ __forceinline__ void set_rr(int c, f64x2 v) { rr_[c] = v; }
    __device__ __forceinline__ f64x2 dd(int c) const { return dd_[c]; }
    __device__ __forceinline__ void set_dd(int c, f64x2 v) { dd_[c] = v; }
    __device__ __forceinline__ double h(int c) const { return h_[c]; }
    __device__ __forceinline__ void set_h(int c, double v) { h_[c] = v; }
    __device__ __forceinline__ double a(int c) const { return a_[c]; }
    __device__ __forceinline__ void set_a(int c, double v) { a_[c] = v; }
    __device__ __forceinline__ double t(int c) const { return t_[c]; }
    __device__ __forceinline__ void set_t(int c, double v) { t_[c] = v; }
    __device__ __forceinline__ double w(int c) const { return w_[c]; }
    __device__ __forceinline__ void set_w(int c, double v) { w_[c] = v; }
    __device__ __forceinline__ f64x2 x(int a) const { return x_[a]; }
    __device__ __forceinline__ void set_x(int a, f64x2 v) { x_[a] = v; }
    __device__ __forceinline__ f64x2 dx(int a) const { return dx_[a]; }
    __device__ __forceinline__ void set_dx(int a, f64x2 v) { dx_[a] = v; }
};

// K <- 2I + sum_c w_c a_c a_c'  (a_c = -2 e_c at robot i, +2 e_c at robot j; w4 = 4 w): diagonal blocks sum over the partners in
// row order, an off-diagonal block is -w4 e e' of its one pair.  Rows are read from the workspace.
template <int N, typename Rows>
__device__ __forceinline__ void assemble(const Rows &R, double (&K)[N * (2 * N + 1)], bool unit_weights) {
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
            const f64x2 e = R.e(c);
            const double w4 = unit_weights ? 4.0 : R.w(c);
            const double a = w4 * e.x, b = w4 * e.y;
            const double wxx = a * e.x, wxy = a * e.y, wyy = b * e.y;
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
        rd[j] = rcp_spec(d);
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

__device__ __forceinline__ void phase_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// The step to the boundary t = max(0, -ds/s, -dz/z) as a fraction n / d (d > 0), compared by cross-multiplication.  Order of the
// comparisons (oracle_core.h): LANES interleaved runs of rows, each folded in row order from 0 / 1, then a balanced tree in which
// the lower run is kept unless the higher one is strictly larger.
struct Frac {
    double n, d;
};
__device__ __forceinline__ Frac frac_fold(Frac a, double n, double d) { return (n * a.d > a.n * d) ? Frac{n, d} : a; }
__device__ __forceinline__ Frac frac_meet(Frac lo, Frac hi) { return (hi.n * lo.d > lo.n * hi.d) ? hi : lo; }
template <int K>
__device__ __forceinline__ double xor_lane_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = static_cast<unsigned>(xor_lane_i<K>(static_cast<int>(b))), hi = static_cast<unsigned>(xor_lane_i<K>(static_cast<int>(b >> 32)));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
// the tree over the 8 lanes of a group (every lane ends with the same fraction): stage w pairs lane l with l ^ w
template <int W>
__device__ __forceinline__ Frac frac_meet_lanes(Frac mine, int sub) {
    const Frac other = {xor_lane_f64<W>(mine.n), xor_lane_f64<W>(mine.d)};
    return (sub & W) ? frac_meet(other, mine) : frac_meet(mine, other);
}

// the rows of lane `sub`: f(row c, i, j) for c = sub, sub + GS, ... < m.  Every lane of the wavefront executes the same
// instructions; the row (and its robots) differ per lane, so per-robot data is fetched from LDS by address.  The slot loop is
// unrolled at compile time (the slot's packed (i, j) table is a constant).
template <int N, int GS, int Q, typename F>
__device__ __forceinline__ void my_rows_from(int sub, F &f) {
    constexpr int m = N * (N - 1) / 2, SLOTS = (m + GS - 1) / GS;
    if constexpr (Q < SLOTS) {
        if constexpr (GS == 1) {
            constexpr int I = row_i(N, Q), J = row_j(N, Q);
            f(Q, I, J);
        } else {
            constexpr unsigned long long TAB = slot_table<N, GS>(Q);
            const int c = Q * GS + sub;
            const unsigned ij = static_cast<unsigned>(TAB >> (6 * sub)) & 63u;
            if constexpr ((Q + 1) * GS <= m) {
                f(c, static_cast<int>(ij & 7u), static_cast<int>(ij >> 3));
            } else {
                if (c < m) f(c, static_cast<int>(ij & 7u), static_cast<int>(ij >> 3));
            }
        }
        my_rows_from<N, GS, Q + 1>(sub, f);
    }
}
template <int N, int GS, typename F>
__device__ __forceinline__ void my_rows(int sub, F &&f) {
    my_rows_from<N, GS, 0>(sub, f);
}

// io: N records (xi_x, xi_y, uhat_x, uhat_y) of the env (LDS), uhat already thresholded to the magnitude limit; the iterate
// replaces uhat.  ws: the env's workspace, ws_doubles(N) doubles of LDS, 16-byte aligned.  sub: this lane's index in the group of
// GS lanes that share the env (every lane of the group must call, with the same arguments).  Returns cvxopt's `iterations`.
// Register budget: callers are one-wave workgroups with __launch_bounds__(64); the compiler propagates their budget (512
// registers per lane incl. the accumulation registers) to this out-of-line body.
template <int N, int GS>
__device__ __attribute__((noinline)) int solve_qp(const Consts k, float4 *io_generic, double *ws_generic, int sub) {
    static_assert(N >= 2 && N <= MAX_N, "agent count");
    constexpr int n = 2 * N, m = N * (N - 1) / 2;
    // rows in registers when every lane computes every row anyway and they fit (GS = 1), in the env's LDS workspace otherwise
    typedef typename std::conditional<GS == 1, RegRows<N>, LdsRows<N>>::type Rows;
    Rows R;
    if constexpr (GS != 1) R.ws = (lds_f64 *)ws_generic;
    lds_f32x4 *io = (lds_f32x4 *)io_generic;
    double q[n], x[n], rx[n], dx[n], rd[n], K[N * (2 * N + 1)];
    double nq = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        const f32x4 r = io[a];
        q[2 * a] = -2.0 * static_cast<double>(r.z);
        q[2 * a + 1] = -2.0 * static_cast<double>(r.w);
        nq = __builtin_fma(q[2 * a], q[2 * a], nq);
        nq = __builtin_fma(q[2 * a + 1], q[2 * a + 1], nq);
    }
    // rows: e = xi_i - xi_j, h = gamma (|e|^2 - r^2)^3   [distributed]
    my_rows<N, GS>(sub, [&](int c, int i, int j) {
        const f32x4 ri = io[i], rj = io[j];
        const double ex = static_cast<double>(ri.x) - static_cast<double>(rj.x), ey = static_cast<double>(ri.y) - static_cast<double>(rj.y);
        const double hh = __builtin_fma(ex, ex, ey * ey) - k.r2;
        const double gain = (hh >= 0.0 || !k.has_unsafe) ? k.gain : k.ugain;
        R.set_e(c, f64x2{ex, ey});
        R.set_h(c, gain * ((hh * hh) * hh));
    });
    phase_fence();
    double nh = 0.0;
#pragma unroll
    for (int c = 0; c < m; ++c) {
        const double h = R.h(c);
        nh = __builtin_fma(h, h, nh);
    }
    const double resx0sq = k.feas2 * (nq > 1.0 ? nq : 1.0), resz0sq = k.feas2 * (nh > 1.0 ? nh : 1.0);
    // default starting point: (2I + G'G) x = -q + G'h;  z = G x - h;  s = -z;  both shifted into the cone if they are not inside
    assemble<N>(R, K, true);
    ldl<N>(K, rd);
#pragma unroll
    for (int kk = 0; kk < n; ++kk) x[kk] = -q[kk];
    {
        int c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; ++i)
#pragma unroll
            for (int j = i + 1; j < N; ++j, ++c) {
                const f64x2 e = R.e(c);
                const double t = 2.0 * R.h(c);
                x[2 * i] = __builtin_fma(-t, e.x, x[2 * i]);
                x[2 * i + 1] = __builtin_fma(-t, e.y, x[2 * i + 1]);
                x[2 * j] = __builtin_fma(t, e.x, x[2 * j]);
                x[2 * j + 1] = __builtin_fma(t, e.y, x[2 * j + 1]);
            }
    }
    solve<N>(K, rd, x);
#pragma unroll
    for (int a = 0; a < N; ++a) R.set_x(a, f64x2{x[2 * a], x[2 * a + 1]});
    phase_fence();
    my_rows<N, GS>(sub, [&](int c, int i, int j) {   // z = G x - h, s = -z   [distributed]
        const f64x2 e = R.e(c);
        const f64x2 xi = R.x(i), xj = R.x(j);
        const double gx_ = 2.0 * __builtin_fma(e.x, xj.x - xi.x, e.y * (xj.y - xi.y));
        const double z = gx_ - R.h(c);
        R.set_sz(c, f64x2{-z, z});
    });
    phase_fence();
    {
        double ns = 0.0, tz = -1e300, ts = -1e300;
#pragma unroll
        for (int c = 0; c < m; ++c) {
            const f64x2 sz = R.sz(c);
            ns = __builtin_fma(sz.y, sz.y, ns);
            ts = sz.y > ts ? sz.y : ts;  // max(-s)
            tz = sz.x > tz ? sz.x : tz;  // max(-z)
        }
        // t >= -1e-8 max(|s|, 1), without the root: t >= 0, or t^2 <= 1e-16 max(s.s, 1)
        const double lim2 = 1e-16 * (ns > 1.0 ? ns : 1.0);
        const bool shift_s = ts >= 0.0 || ts * ts <= lim2, shift_z = tz >= 0.0 || tz * tz <= lim2;
        const double as = 1.0 + ts, az = 1.0 + tz;
#pragma unroll
        for (int c = 0; c < m; ++c) {   // (replicated: every lane of the group writes the same values)
            f64x2 sz = R.sz(c);
            sz.x = shift_s ? sz.x + as : sz.x;
            sz.y = shift_z ? sz.y + az : sz.y;
            R.set_sz(c, sz);
        }
    }
    // Per row, everything an iteration needs from (x, s, z) in ONE distributed phase: rz = s - h + G x, t = 2 z (the row factor of
    // G'z), 1 / s and the KKT weight 4 z / s.  (x is published; the last two go unused in the iteration that stops.)
    auto row_phase = [&](bool update, double step) {
        my_rows<N, GS>(sub, [&](int c, int i, int j) {
            const f64x2 e = R.e(c);
            f64x2 sz = R.sz(c);
            if (update) {   // s, z <- s + step ds, z + step dz
                const f64x2 d = R.dd(c);
                sz.x = __builtin_fma(step, d.x, sz.x);
                sz.y = __builtin_fma(step, d.y, sz.y);
                R.set_sz(c, sz);
            }
            const f64x2 xi = R.x(i), xj = R.x(j);
            const double gx_ = 2.0 * __builtin_fma(e.x, xj.x - xi.x, e.y * (xj.y - xi.y));
            const double rs = rcp_spec(sz.x);
            R.set_rr(c, f64x2{(sz.x - R.h(c)) + gx_, rs});
            R.set_t(c, 2.0 * sz.y);
            R.set_w(c, 4.0 * (sz.y * rs));
        });
    };
    phase_fence();
    row_phase(false, 0.0);
    int iters = 0;
    RG_IPM_T0()
    for (;; ++iters) {
        phase_fence();
        RG_IPM_TICK(0)
        // residuals rx = q + 2x + G'z, costs, the gap s.z and the sums of the stopping rule   [replicated, row order]
        double f0 = 0.0, nrx = 0.0, nrz = 0.0, zrz = 0.0, gap = 0.0;
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
                    const f64x2 e = R.e(c);
                    const double t = R.t(c), rz = R.rz(c);
                    rx[2 * i] = __builtin_fma(-t, e.x, rx[2 * i]);
                    rx[2 * i + 1] = __builtin_fma(-t, e.y, rx[2 * i + 1]);
                    rx[2 * j] = __builtin_fma(t, e.x, rx[2 * j]);
                    rx[2 * j + 1] = __builtin_fma(t, e.y, rx[2 * j + 1]);
                    nrz = __builtin_fma(rz, rz, nrz);
                    zrz = __builtin_fma(0.5 * t, rz, zrz);   // z = t / 2 exactly
                    gap = __builtin_fma(R.s(c), 0.5 * t, gap);
                }
        }
#pragma unroll
        for (int kk = 0; kk < n; ++kk) nrx = __builtin_fma(rx[kk], rx[kk], nrx);
        const double pcost = f0, dcost = (f0 + zrz) - gap;
        const bool rel_ok = pcost < 0.0 ? gap <= k.reltol * -pcost : dcost > 0.0 ? gap <= k.reltol * dcost : false;
        if ((nrz <= resz0sq && nrx <= resx0sq && (gap <= k.abstol || rel_ok)) || iters == k.maxiters) break;
        RG_IPM_TICK(1)
        assemble<N>(R, K, false);
        RG_IPM_TICK(2)
        ldl<N>(K, rd);
        RG_IPM_TICK(3)
        const double mu = gap * (1.0 / static_cast<double>(m));
        double sigmamu = 0.0, step = 1.0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            // the row factor of the right-hand side: t = 2 (rc + z rz) / s,  rc = -s z + sigma mu [- dsa dza]   [distributed]
            my_rows<N, GS>(sub, [&](int c, int, int) {
                const f64x2 sz = R.sz(c), rr = R.rr(c);
                const double rc = pass ? __builtin_fma(-sz.x, sz.y, sigmamu) - R.a(c) : -(sz.x * sz.y);
                R.set_t(c, 2.0 * (__builtin_fma(sz.y, rr.x, rc) * rr.y));
            });
            phase_fence();
#pragma unroll
            for (int kk = 0; kk < n; ++kk) dx[kk] = -rx[kk];
            {
                int c = 0;
#pragma unroll
                for (int i = 0; i < N - 1; ++i)
#pragma unroll
                    for (int j = i + 1; j < N; ++j, ++c) {
                        const f64x2 e = R.e(c);
                        const double t = R.t(c);
                        dx[2 * i] = __builtin_fma(t, e.x, dx[2 * i]);
                        dx[2 * i + 1] = __builtin_fma(t, e.y, dx[2 * i + 1]);
                        dx[2 * j] = __builtin_fma(-t, e.x, dx[2 * j]);
                        dx[2 * j + 1] = __builtin_fma(-t, e.y, dx[2 * j + 1]);
                    }
            }
            solve<N>(K, rd, dx);
#pragma unroll
            for (int a = 0; a < N; ++a) R.set_dx(a, f64x2{dx[2 * a], dx[2 * a + 1]});
            phase_fence();
            RG_IPM_TICK(4)
            // ds = -rz - G dx;  dz = (rc - z ds) / s;  the predictor also leaves dsa dza for the corrector   [distributed]; each lane
            // folds its own rows' candidates for the step to the boundary as it goes
            Frac best = {0.0, 1.0};
            my_rows<N, GS>(sub, [&](int c, int i, int j) {
                const f64x2 e = R.e(c), sz = R.sz(c);
                const f64x2 rr = R.rr(c);
                const f64x2 di = R.dx(i), dj = R.dx(j);
                const double gdx = 2.0 * __builtin_fma(e.x, dj.x - di.x, e.y * (dj.y - di.y));
                const double rc = pass ? __builtin_fma(-sz.x, sz.y, sigmamu) - R.a(c) : -(sz.x * sz.y);
                const double ds = -rr.x - gdx;
                const double dz = __builtin_fma(-sz.y, ds, rc) * rr.y;
                R.set_dd(c, f64x2{ds, dz});
                if (pass == 0) R.set_a(c, ds * dz);
                if constexpr (GS == 8) {
                    best = frac_fold(best, -ds, sz.x);
                    best = frac_fold(best, -dz, sz.y);
                }
            });
            phase_fence();
            RG_IPM_TICK(5)
            // sum ds dz [replicated, row order] and the step to the boundary as a fraction tn / td
            double dsdz = 0.0, tn, td;
#pragma unroll
            for (int c = 0; c < m; ++c) {
                const f64x2 d = R.dd(c);
                dsdz = __builtin_fma(d.x, d.y, dsdz);
            }
            if constexpr (GS == 8) {   // N >= 5: the spec's 8 interleaved runs are the group's lanes; they meet by lane permutes
                static_assert(N >= 5, "groups of 8 carry five to eight robots");
                best = frac_meet_lanes<1>(best, sub);
                best = frac_meet_lanes<2>(best, sub);
                best = frac_meet_lanes<4>(best, sub);
                tn = best.n;
                td = best.d;
            } else {                   // every row in this lane
                constexpr int LANES = N <= 4 ? 1 : 8;
                Frac p[LANES];
#pragma unroll
                for (int l = 0; l < LANES; ++l) {
                    p[l] = Frac{0.0, 1.0};
#pragma unroll
                    for (int c = l; c < m; c += LANES) {
                        const f64x2 d = R.dd(c), sz = R.sz(c);
                        p[l] = frac_fold(p[l], -d.x, sz.x);
                        p[l] = frac_fold(p[l], -d.y, sz.y);
                    }
                }
#pragma unroll
                for (int w = 1; w < LANES; w *= 2)
#pragma unroll
                    for (int l = 0; l < LANES; l += 2 * w) p[l] = frac_meet(p[l], p[l + w]);
                tn = p[0].n;
                td = p[0].d;
            }
            if (pass == 0) {
                step = tn > td ? td * rcp_spec(tn) : 1.0;
                double sg = __builtin_fma(dsdz * rcp_spec(gap), step * step, 1.0 - step);
                sg = sg < 0.0 ? 0.0 : sg > 1.0 ? 1.0 : sg;
                sigmamu = ((sg * sg) * sg) * mu;
            } else {
                step = 0.99 * td < tn ? (0.99 * td) * rcp_spec(tn) : 1.0;
            }
            RG_IPM_TICK(6)
        }
#pragma unroll
        for (int kk = 0; kk < n; ++kk) x[kk] = __builtin_fma(step, dx[kk], x[kk]);
#pragma unroll
        for (int a = 0; a < N; ++a) R.set_x(a, f64x2{x[2 * a], x[2 * a + 1]});
        phase_fence();
        row_phase(true, step);   // s, z updated; the next iteration's row data
        RG_IPM_TICK(7)
    }
    phase_fence();
#pragma unroll
    for (int a = 0; a < N; ++a) {   // (every lane of the group writes the same values)
        reinterpret_cast<lds_f32x2 *>(io + a)[1] = f32x2{static_cast<float>(x[2 * a]), static_cast<float>(x[2 * a + 1])};
    }
    return iters;
}

// runtime agent count -> the out-of-line body (N = 1: no rows, the unconstrained minimiser is the thresholded input itself).
// GW: the caller's lane-group width.  N <= 4 (groups of 4; also one lane per env, N <= 5): every lane runs the whole iteration
// with the rows in registers -- measured faster than sharing them through LDS up to N = 4 (tests/ipm_bench.py: 5.5 k
// against 7.4 k cycles per iteration at N = 4); N = 5 .. 8 (groups of 8): rows in LDS, row phases spread over the 8 lanes.
template <int GW>
__device__ __forceinline__ int solve_qp_n(int N, const Consts &k, float4 *io, double *ws, int sub) {
    if constexpr (GW == 8) {
        switch (N) {
            case 5: return solve_qp<5, 8>(k, io, ws, sub);
            case 6: return solve_qp<6, 8>(k, io, ws, sub);
            case 7: return solve_qp<7, 8>(k, io, ws, sub);
            case 8: return solve_qp<8, 8>(k, io, ws, sub);
            default: return 0;
        }
    } else {
        switch (N) {
            case 2: return solve_qp<2, 1>(k, io, ws, 0);
            case 3: return solve_qp<3, 1>(k, io, ws, 0);
            case 4: return solve_qp<4, 1>(k, io, ws, 0);
            case 5: if constexpr (GW == 1) return solve_qp<5, 1>(k, io, ws, 0); else return 0;
            default: return 0;
        }
    }
}

// LDS of the interior-point mode for a wavefront whose lane groups are GW wide: the records and one workspace per env
template <int GW>
struct alignas(16) GroupLds {
    float4 rec[64];
    double ws[64 / GW][GW == 8 ? ws_doubles(8) : 2];   // (groups of 4: the rows live in registers)
};

}  // namespace ipm
}  // namespace rg
