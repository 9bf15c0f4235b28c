/* ORACLE -- test infrastructure only.  Never linked, imported or executed by marbler_amd/.
 *
 * CPU restatement ("sim_spec_v0") of the one hot path this repo accelerates: one
 * Robotarium-gym env step = goal generation -> U sim sub-iterations {controller every 15th:
 * uni->si, position controller, barrier-certificate QP, si->uni; validation; Euler
 * integration} -> scenario tracking / observation / reward / termination.
 *
 * This file is included twice by oracle.c: once with REAL=double and libm (tier 2: follows
 * the reference + rps operation for operation, checked against tests/golden/), once with
 * REAL=float and the spec'd polynomial sin/cos/atan2 (tier 3: the bit-exact twin of the HIP
 * kernels, which implement the same arithmetic in the same order on lane groups).
 *
 * The arithmetic below the reference's own layers lives in the third-party package rps
 * (robotarium_python_simulator, pinned by prose to commit 6bb184e, README.md:11) and cvxopt,
 * both absent from /root/reference and from this image: rows a4-a10 are restated from the
 * spec in SURVEY.md Appendix A and are PARITY UNPINNED against real rps + cvxopt.  Rows a1,
 * a2, a11-a16 are pinned by tests/golden/ npz files (captured from the reference's own Python).
 *
 * Reference citations are relative to /root/reference/robotarium_gym/.
 */

#ifndef REAL
#error "include from oracle.c"
#endif

/* ---------------------------------------------------------------- math per precision */
#if ORC_IS_F32
/* sim_spec_v0 float math: every operation is an IEEE-754 binary32 +,-,*,/,sqrt or an explicit
 * fma; no contraction (compile with -ffp-contract=off).  The HIP kernels use the same
 * sequence, so results are bit-identical. */
static inline void FN(sincos)(float t, float *s, float *c) {
    float q = rintf(t * 0.63661977236758134f);
    float r = __builtin_fmaf(q, -1.57079625129699707031f, t);     /* pi/2 hi */
    r = __builtin_fmaf(q, -7.54978941586159635335e-08f, r);       /* pi/2 lo */
    float z = r * r;
    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(z, sp, -1.6666654611e-1f);
    float sr = __builtin_fmaf(r * z, sp, r);
    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(z, cp, 4.166664568298827e-2f);
    float cr = __builtin_fmaf(z * z, cp, __builtin_fmaf(z, -0.5f, 1.0f));
    int k = ((int)q) & 3;
    float ss = (k & 1) ? cr : sr;
    float cc = (k & 1) ? sr : cr;
    if (k == 1 || k == 2) cc = -cc;
    if (k >= 2) ss = -ss;
    *s = ss;
    *c = cc;
}
static inline float FN(atan2)(float y, float x) {
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float a;
    if (mx == 0.0f) {
        a = 0.0f;
    } else {
        /* atan(mn/mx) on [0,1]; above tan(pi/8) use atan(t) = pi/4 + atan((t-1)/(t+1)) */
        int hi = mn > 0.4142135679721832275390625f * mx;
        float num = hi ? (mn - mx) : mn;
        float den = hi ? (mn + mx) : mx;
        float t = num / den;
        float z = t * t;
        float p = __builtin_fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
        p = __builtin_fmaf(z, p, 1.99777106478e-1f);
        p = __builtin_fmaf(z, p, -3.33329491539e-1f);
        a = __builtin_fmaf(p * z, t, t);
        if (hi) a = a + 0.785398185253143310546875f;
    }
    if (ay > ax) a = 1.57079637050628662109375f - a;
    if (x < 0.0f) a = 3.1415927410125732421875f - a;
    return (y < 0.0f) ? -a : a;
}
/* rps wraps with theta = atan2(sin theta, cos theta) (Appendix A.4): the identity on (-pi, pi].
 * In binary32 that round trip perturbs theta by ~1e-7 every sub-iteration; the float spec
 * instead subtracts 2*pi (hi + lo) only when |theta| exceeds pi -- closer to the float64
 * sequence it stands for (measured: tests/test_oracle_spec.py) and free of transcendentals. */
static inline float FN(wrap)(float t) {
    if (t > 3.1415927410125732421875f) t = (t - 6.283185482025146484375f) - (-1.74845553146951715462e-07f);
    else if (t < -3.1415927410125732421875f) t = (t + 6.283185482025146484375f) + (-1.74845553146951715462e-07f);
    return t;
}
/* sin/cos of the per-sub-step heading increment dt*w (|dt*w| <= 0.033 * 3.64 = 0.12 rad with the rps
 * constants): Taylor polynomials without range reduction (truncation < 1e-10 for |t| <= 0.25); larger
 * arguments take the general routine. */
static inline void FN(sincos_step)(float t, float *s, float *c) {
    if (__builtin_fabsf(t) <= 0.25f) {
        float z = t * t;
        float sp = __builtin_fmaf(z, 8.33333377e-3f, -1.66666672e-1f);
        *s = __builtin_fmaf(t * z, sp, t);
        float cp = __builtin_fmaf(z, -1.38888892e-3f, 4.16666679e-2f);
        cp = __builtin_fmaf(z, cp, -0.5f);
        *c = __builtin_fmaf(z, cp, 1.0f);
    } else {
        FN(sincos)(t, s, c);
    }
}
#define SINCOS_STEP(t, s, c) FN(sincos_step)((t), (s), (c))
#define WRAP(t) FN(wrap)(t)
#define SINCOS(t, s, c) FN(sincos)((t), (s), (c))
#define ATAN2(y, x) FN(atan2)((y), (x))
#define SQRT(v) __builtin_sqrtf(v)
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
/* float spec: range tests and nearest-neighbour / nearest-prey ordering compare SQUARED distances
 * (d2 <= r*r, d2_a < d2_b): the same decisions as the reference's norm() <= r except within an ulp
 * of the threshold, and no sqrt in the scenario epilogue */
#define DIST(d2) (d2)
#define RADIUS(r) ((r) * (r))
#else
#define SINCOS(t, s, c) do { *(s) = sin(t); *(c) = cos(t); } while (0)
#define ATAN2(y, x) atan2((y), (x))
#define WRAP(t) atan2(sin(t), cos(t))
#define SQRT(v) sqrt(v)
#define FMA(a, b, c) __builtin_fma((a), (b), (c))
#define DIST(d2) sqrt(d2)
#define RADIUS(r) (r)
#endif

#define R(v) ((REAL)(v))

typedef struct {
    REAL *poses;            /* [E][3][N]  (x row, y row, theta row: the reference's 3xN, roboEnv.py:54) */
    REAL *carry;            /* [E][N] movement of the last sub-iteration of the previous step, not yet
                               counted in dist_travelled (roboEnv.py:55-59: previous_pose lags one iteration) */
    int32_t *steps;         /* [E] episode_steps */
    REAL *prey_loc;         /* [E][P][2]  PCP */
    uint8_t *prey_sensed;   /* [E][P] */
    uint8_t *prey_captured; /* [E][P] */
    uint8_t *loaded;        /* [E][N]     Warehouse */
    int32_t *load;          /* [E][N]     MaterialTransport */
    int32_t *zone_load;     /* [E][2] */
    int32_t *messages;      /* [E][4] */
    uint8_t *grid;          /* [E][96]    ArcticTransport */
    int32_t *goal_col;      /* [E] */
    uint8_t *pixel_type;    /* [E][N] */
    uint8_t *reached_goal;  /* [E][N] */
} FN(orc_state);

typedef struct {
    REAL *obs;          /* [E][N][D] */
    REAL *reward;       /* [E][N] */
    uint8_t *done;      /* [E] */
    REAL *dist;         /* [E][N] */
    uint8_t *viol;      /* [E] 0 '', 1 collision, 2 boundary, 3 collision_boundary */
    int32_t *remaining; /* [E] -1 when the reference's info has no 'remaining' */
    int32_t *qp_sweeps; /* [E] max sweeps of any QP of this step (diagnostic), may be NULL */
} FN(orc_out);

static inline REAL FN(clampv)(REAL v, REAL lo, REAL hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* a6: barrier certificate as an exact projection (Hildreth sweeps in XOR-factorisation order) */
static int FN(barrier_qp)(const orc_params *p, int N, const REAL *xix, const REAL *xiy, REAL *ux, REAL *uy) {
    /* min ||u - uhat||^2  s.t.  e_ij.(u_j - u_i) <= beta_ij  (rps' rows halved; e_ij = xi_i - xi_j,
     * beta = gamma h^3 / 2), solved as the exact projection by Hildreth's dual coordinate ascent:
     * multipliers mu_ij >= 0, u = uhat + sum_ij mu_ij (e_ij at i, -e_ij at j).  Pairs are visited in
     * the XOR 1-factorisation order (k = 1..GW-1, pairs (a, a^k)) that the HIP kernel runs
     * pair-parallel.  One pair update, with f = e/n2, bp = beta/n2, n2 = 2|e|^2:
     *     mn = max(0, mu - bp + f.(u_j - u_i));  delta = mn - mu;  u_i += delta e;  u_j -= delta e.
     * Coupled constraints converge geometrically (ratio <= 1/4 per sweep for two pairs sharing a
     * robot, closer to 1 for clusters of three and more), so after sweeps 3, 7, 11, ... the whole
     * multiplier vector is extrapolated along its last change: with d1, d2 the last two changes of mu and
     * dd = d2 - d1, gamma = <dd, d2> / <dd, dd> (the least-squares fit of d2 ~ r d1, gamma = r / (r - 1)) and
     * mu <- max(0, mu - gamma d2) -- Aitken's delta-squared for vectors (reduced-rank extrapolation of order one):
     * ONE division per restart, accepted for 0 < r < 32/33 -- and u is rebuilt from the multipliers: a restart
     * from a better point of the same convergent iteration.  (Rounds 1-3 extrapolated each multiplier by its
     * own scalar Aitken step: one division per pair, 1.7 sweeps' worth of time per restart on the GPU, and a
     * longer tail -- DESIGN.md section 4.4.)  The two inner products are summed per robot over its partners
     * in round order and then over the robots in the order of the HIP kernel's lane butterfly. */
    int gw = 2;
    while (gw < N) gw *= 2;
    REAL r2 = R(p->safety_radius) * R(p->safety_radius);
    REAL ex[ORC_MAXN][ORC_MAXN], ey[ORC_MAXN][ORC_MAXN], fx[ORC_MAXN][ORC_MAXN], fy[ORC_MAXN][ORC_MAXN],
        bp[ORC_MAXN][ORC_MAXN], emax[ORC_MAXN][ORC_MAXN], mu[ORC_MAXN][ORC_MAXN], muA[ORC_MAXN][ORC_MAXN],
        muB[ORC_MAXN][ORC_MAXN];
    int valid[ORC_MAXN][ORC_MAXN];
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            REAL dx = xix[i] - xix[j], dy = xiy[i] - xiy[j];
            REAL ee = dx * dx + dy * dy;
            REAL h = ee - r2;
            REAL gain = (h >= R(0) || !p->barrier_has_unsafe_gain) ? R(p->barrier_gain) : R(p->unsafe_barrier_gain);
            REAL b = gain * ((h * h) * h);
            REAL n2 = R(2) * ee;
            valid[i][j] = n2 > R(0);
            REAL rn2 = valid[i][j] ? R(1) / n2 : R(0); /* one division per pair per QP */
            REAL ax = dx < R(0) ? -dx : dx, ay = dy < R(0) ? -dy : dy;
            ex[i][j] = dx;
            ey[i][j] = dy;
            fx[i][j] = dx * rn2;
            fy[i][j] = dy * rn2;
            bp[i][j] = (R(0.5) * b) * rn2;
            emax[i][j] = ax > ay ? ax : ay;
            mu[i][j] = muA[i][j] = muB[i][j] = R(0);
        }
    /* "Threshold control inputs before QP" */
    REAL uhx[ORC_MAXN], uhy[ORC_MAXN];
    for (int a = 0; a < N; ++a) {
#if ORC_IS_F32
        /* float spec: the decision on squares (after the 0.15 position clip this branch is never
         * taken with the rps constants, so the common path has no sqrt / divide) */
        REAL n2u = ux[a] * ux[a] + uy[a] * uy[a];
        if (n2u > R(p->barrier_magnitude_limit) * R(p->barrier_magnitude_limit)) {
            REAL sc = R(p->barrier_magnitude_limit) / SQRT(n2u);
#else
        REAL nrm = SQRT(ux[a] * ux[a] + uy[a] * uy[a]);
        if (nrm > R(p->barrier_magnitude_limit)) {
            REAL sc = R(p->barrier_magnitude_limit) / nrm;
#endif
            ux[a] = ux[a] * sc;
            uy[a] = uy[a] * sc;
        }
        uhx[a] = ux[a];
        uhy[a] = uy[a];
    }
    int sweeps = 0;
    for (;;) {
        REAL maxchg = R(0);
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j) {
                muA[i][j] = muB[i][j];
                muB[i][j] = mu[i][j];
            }
        for (int k = 1; k < gw; ++k)
            for (int i = 0; i < N; ++i) {
                int j = i ^ k;
                if (!(i < j && j < N) || !valid[i][j]) continue;
                REAL c0 = mu[i][j] - bp[i][j];
                REAL t = FMA(fy[i][j], uy[j] - uy[i], c0);
                REAL mn = FMA(fx[i][j], ux[j] - ux[i], t);
                if (!(mn > R(0))) mn = R(0);
                REAL delta = mn - mu[i][j];
                mu[i][j] = mn;
                ux[i] = FMA(delta, ex[i][j], ux[i]);
                uy[i] = FMA(delta, ey[i][j], uy[i]);
                ux[j] = FMA(-delta, ex[i][j], ux[j]);
                uy[j] = FMA(-delta, ey[i][j], uy[j]);
                REAL chg = (delta < R(0) ? -delta : delta) * emax[i][j]; /* = max(|delta ex|, |delta ey|) */
                if (chg > maxchg) maxchg = chg;
            }
        ++sweeps;
        /* converged when the largest component change of the sweep is below qp_rtol relative to
         * max(|u|_inf, magnitude_limit) */
        REAL umax = R(p->barrier_magnitude_limit);
        for (int a = 0; a < N; ++a) {
            REAL ax = ux[a] < R(0) ? -ux[a] : ux[a], ay = uy[a] < R(0) ? -uy[a] : uy[a];
            if (ax > umax) umax = ax;
            if (ay > umax) umax = ay;
        }
        if (!(maxchg > R(p->qp_rtol) * umax) || sweeps >= p->qp_max_sweeps) break;
        if ((sweeps & 3) == 3) { /* restart: one extrapolation factor for the whole multiplier vector */
            REAL pa[ORC_MAXN], pb[ORC_MAXN];
            for (int a = 0; a < gw; ++a) { /* per robot, over its partners in round order (each pair counts twice) */
                pa[a] = pb[a] = R(0);
                if (a >= N) continue;
                for (int k = 1; k < gw; ++k) {
                    int q = a ^ k;
                    if (q >= N) continue;
                    int lo = a < q ? a : q, hi = a < q ? q : a;
                    if (!valid[lo][hi]) continue;
                    REAL d1 = muB[lo][hi] - muA[lo][hi], d2 = mu[lo][hi] - muB[lo][hi];
                    REAL dd = d2 - d1;
                    pa[a] = FMA(dd, d2, pa[a]);
                    pb[a] = FMA(dd, dd, pb[a]);
                }
            }
            for (int stride = 1; stride < gw; stride *= 2) /* the lane butterfly of the HIP kernel: (s0+s1)+(s2+s3) ... */
                for (int a = 0; a < gw; a += 2 * stride) {
                    pa[a] = pa[a] + pa[a + stride];
                    pb[a] = pb[a] + pb[a + stride];
                }
            REAL ga = pa[0], gb = pb[0];
            int ok = gb > R(0) && ga < R(0) && -ga < R(32) * gb;
            REAL gam = ok ? ga / gb : R(0); /* the one division of a restart */
            for (int i = 0; i < N; ++i)
                for (int j = i + 1; j < N; ++j) {
                    if (!valid[i][j]) continue;
                    REAL d2 = mu[i][j] - muB[i][j];
                    REAL m = FMA(-gam, d2, mu[i][j]);
                    if (!(m > R(0))) m = R(0);
                    mu[i][j] = m;
                }
            for (int a = 0; a < N; ++a) { /* u = uhat + sum over partners, in round order */
                REAL sx = uhx[a], sy = uhy[a];
                for (int k = 1; k < gw; ++k) {
                    int q = a ^ k;
                    if (q >= N) continue;
                    int lo = a < q ? a : q, hi = a < q ? q : a;
                    if (!valid[lo][hi]) continue;
                    REAL sgn_ex = a < q ? ex[lo][hi] : -ex[lo][hi], sgn_ey = a < q ? ey[lo][hi] : -ey[lo][hi];
                    sx = FMA(mu[lo][hi], sgn_ex, sx);
                    sy = FMA(mu[lo][hi], sgn_ey, sy);
                }
                ux[a] = sx;
                uy[a] = sy;
            }
        }
    }
    return sweeps;
}

#if !ORC_IS_F32
/* STUDY MODE (qp_mode == 1, float64 tier only; never part of a parity claim): the barrier QP solved the way the
 * reference's rps hands it to cvxopt -- `solvers.qp(H, f, A, b)` with H = 2I, f = -2 vec(uhat), rows
 * A[c, 2i:2i+2] = -2 e_ij, A[c, 2j:2j+2] = +2 e_ij, b[c] = gamma h^3 and
 * `options['reltol'] = options['feastol'] = 1e-2, maxiters = 50` -- by a restatement, FROM MEMORY, of cvxopt's coneqp for
 * the linear cone (cvxopt is absent from this image: unpinned): default starting point from the least-squares KKT
 * system, Mehrotra predictor-corrector with Nesterov-Todd scaling (for the linear cone: the plain s, z iteration),
 * step 0.99 to the boundary, sigma = (1 - step + dsdz/gap step^2)^3, stop when pres, dres <= feastol and
 * (gap <= abstol or relgap <= reltol).  Its iterate is an APPROXIMATE, strictly interior solution; how far it sits from
 * the exact projection, and what that does to the statistics a trainer sees, is what tests/free_running.py measures
 * with it. */
static int FN(chol_solve)(int n, REAL *M, REAL *b) { /* M (n x n, row-major, SPD) <- its Cholesky factor; b <- M^-1 b */
    for (int j = 0; j < n; ++j) {
        REAL d = M[j * n + j];
        for (int k = 0; k < j; ++k) d -= M[j * n + k] * M[j * n + k];
        if (!(d > R(0))) return -1;
        d = SQRT(d);
        M[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            REAL v = M[i * n + j];
            for (int k = 0; k < j; ++k) v -= M[i * n + k] * M[j * n + k];
            M[i * n + j] = v / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        REAL v = b[i];
        for (int k = 0; k < i; ++k) v -= M[i * n + k] * b[k];
        b[i] = v / M[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        REAL v = b[i];
        for (int k = i + 1; k < n; ++k) v -= M[k * n + i] * b[k];
        b[i] = v / M[i * n + i];
    }
    return 0;
}

#define ORC_MAXV (2 * ORC_MAXN)
#define ORC_MAXC (ORC_MAXN * (ORC_MAXN - 1) / 2)
static int FN(barrier_qp_ipm)(const orc_params *p, int N, const REAL *xix, const REAL *xiy, REAL *ux, REAL *uy) {
    const int n = 2 * N;
    int m = 0;
    static const REAL STEP = 0.99;
    const REAL ABSTOL = p->ipm_abstol, RELTOL = p->ipm_reltol, FEASTOL = p->ipm_feastol;
    const int MAXITERS = p->ipm_maxiters;
    REAL r2 = R(p->safety_radius) * R(p->safety_radius);
    /* rows in rps' order (i < j, i outer); vec(u) column-major: u[2a] = x of robot a, u[2a+1] = y */
    static _Thread_local REAL G[ORC_MAXC][ORC_MAXV];
    REAL h[ORC_MAXC];
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j) {
            REAL dx = xix[i] - xix[j], dy = xiy[i] - xiy[j];
            REAL hh = dx * dx + dy * dy - r2;
            REAL gain = (hh >= 0 || !p->barrier_has_unsafe_gain) ? p->barrier_gain : p->unsafe_barrier_gain;
            for (int k = 0; k < n; ++k) G[m][k] = 0;
            G[m][2 * i] = -2 * dx; G[m][2 * i + 1] = -2 * dy;
            G[m][2 * j] = 2 * dx;  G[m][2 * j + 1] = 2 * dy;
            h[m] = gain * hh * hh * hh;
            ++m;
        }
    REAL q[ORC_MAXV], x[ORC_MAXV], s[ORC_MAXC], z[ORC_MAXC];
    for (int a = 0; a < N; ++a) { /* "Threshold control inputs before QP" (in place, like rps) */
        REAL nrm = SQRT(ux[a] * ux[a] + uy[a] * uy[a]);
        if (nrm > p->barrier_magnitude_limit) {
            ux[a] *= p->barrier_magnitude_limit / nrm;
            uy[a] *= p->barrier_magnitude_limit / nrm;
        }
        q[2 * a] = -2 * ux[a];
        q[2 * a + 1] = -2 * uy[a];
    }
    REAL nq = 0, nh = 0;
    for (int k = 0; k < n; ++k) nq += q[k] * q[k];
    for (int c = 0; c < m; ++c) nh += h[c] * h[c];
    const REAL resx0 = SQRT(nq) > 1 ? SQRT(nq) : 1, resz0 = SQRT(nh) > 1 ? SQRT(nh) : 1;
    REAL K[ORC_MAXV * ORC_MAXV], rhs[ORC_MAXV];
    /* default starting point: [P G'; G -I][x; z] = [-q; h]  =>  (P + G'G) x = -q + G'h, z = G x - h, s = -z */
    for (int a = 0; a < n; ++a) {
        for (int b = 0; b < n; ++b) {
            REAL v = a == b ? 2 : 0;
            for (int c = 0; c < m; ++c) v += G[c][a] * G[c][b];
            K[a * n + b] = v;
        }
        REAL v = -q[a];
        for (int c = 0; c < m; ++c) v += G[c][a] * h[c];
        rhs[a] = v;
    }
    if (FN(chol_solve)(n, K, rhs)) return -1;
    for (int k = 0; k < n; ++k) x[k] = rhs[k];
    REAL ns = 0, ts = -1e300, tz = -1e300;
    for (int c = 0; c < m; ++c) {
        REAL gx_ = 0;
        for (int k = 0; k < n; ++k) gx_ += G[c][k] * x[k];
        z[c] = gx_ - h[c];
        s[c] = -z[c];
        ns += s[c] * s[c];
        if (-s[c] > ts) ts = -s[c];
        if (-z[c] > tz) tz = -z[c];
    }
    ns = SQRT(ns);   /* (= ||z|| as well) */
    if (ts >= -1e-8 * (ns > 1 ? ns : 1)) for (int c = 0; c < m; ++c) s[c] += 1 + ts;
    if (tz >= -1e-8 * (ns > 1 ? ns : 1)) for (int c = 0; c < m; ++c) z[c] += 1 + tz;
    REAL gap = 0;
    for (int c = 0; c < m; ++c) gap += s[c] * z[c];
    int iters;
    for (iters = 0;; ++iters) {
        REAL rx[ORC_MAXV], rz[ORC_MAXC], f0 = 0, resx = 0, resz = 0, zrz = 0;
        for (int k = 0; k < n; ++k) {
            REAL v = q[k] + 2 * x[k];          /* q + P x */
            f0 += 0.5 * (x[k] * v + x[k] * q[k]);
            for (int c = 0; c < m; ++c) v += G[c][k] * z[c];
            rx[k] = v;
            resx += v * v;
        }
        for (int c = 0; c < m; ++c) {
            REAL v = s[c] - h[c];
            for (int k = 0; k < n; ++k) v += G[c][k] * x[k];
            rz[c] = v;
            resz += v * v;
            zrz += z[c] * v;
        }
        resx = SQRT(resx);
        resz = SQRT(resz);
        const REAL pcost = f0, dcost = f0 + zrz - gap;
        const int have_rel = pcost < 0 || dcost > 0;
        const REAL relgap = pcost < 0 ? gap / -pcost : dcost > 0 ? gap / dcost : 0;
        const REAL pres = resz / resz0, dres = resx / resx0;
        if ((pres <= FEASTOL && dres <= FEASTOL && (gap <= ABSTOL || (have_rel && relgap <= RELTOL))) || iters == MAXITERS) break;
        /* Newton systems: (P + G' diag(z/s) G) dx = -rx - G'[(rc + z o rz) / s];  ds = -rz - G dx;  dz = (rc - z o ds) / s */
        REAL Kf[ORC_MAXV * ORC_MAXV];
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                REAL v = a == b ? 2 : 0;
                for (int c = 0; c < m; ++c) v += G[c][a] * (z[c] / s[c]) * G[c][b];
                Kf[a * n + b] = v;
            }
        REAL dx[ORC_MAXV], ds[ORC_MAXC], dz[ORC_MAXC], dsa_dza[ORC_MAXC];
        const REAL mu = gap / m;
        REAL sigma = 0, step = 1;
        int failed = 0;
        for (int pass = 0; pass < 2 && !failed; ++pass) {
            REAL rc[ORC_MAXC], Kc[ORC_MAXV * ORC_MAXV];
            for (int c = 0; c < m; ++c) rc[c] = -s[c] * z[c] + sigma * mu - (pass ? dsa_dza[c] : 0);
            for (int a = 0; a < n; ++a) {
                REAL v = -rx[a];
                for (int c = 0; c < m; ++c) v -= G[c][a] * ((rc[c] + z[c] * rz[c]) / s[c]);
                dx[a] = v;
            }
            for (int k = 0; k < n * n; ++k) Kc[k] = Kf[k];
            if (FN(chol_solve)(n, Kc, dx)) { failed = 1; break; }
            REAL dsdz = 0, t = 0;
            for (int c = 0; c < m; ++c) {
                REAL gdx = 0;
                for (int k = 0; k < n; ++k) gdx += G[c][k] * dx[k];
                ds[c] = -rz[c] - gdx;
                dz[c] = (rc[c] - z[c] * ds[c]) / s[c];
                dsdz += ds[c] * dz[c];
                if (-ds[c] / s[c] > t) t = -ds[c] / s[c];
                if (-dz[c] / z[c] > t) t = -dz[c] / z[c];
            }
            if (pass == 0) {
                step = t == 0 ? 1 : (1 / t < 1 ? 1 / t : 1);
                REAL sg = 1 - step + dsdz / gap * step * step;
                sg = sg < 0 ? 0 : sg > 1 ? 1 : sg;
                sigma = sg * sg * sg;
                for (int c = 0; c < m; ++c) dsa_dza[c] = ds[c] * dz[c];
            } else {
                step = t == 0 ? 1 : (STEP / t < 1 ? STEP / t : 1);
            }
        }
        if (failed) break;
        for (int k = 0; k < n; ++k) x[k] += step * dx[k];
        gap = 0;
        for (int c = 0; c < m; ++c) {
            s[c] += step * ds[c];
            z[c] += step * dz[c];
            gap += s[c] * z[c];
        }
    }
    for (int a = 0; a < N; ++a) {
        ux[a] = x[2 * a];
        uy[a] = x[2 * a + 1];
    }
    return iters;
}
#endif

#if !ORC_IS_F32
/* a6, qp_mode == 1 of the float spec ("ipm_spec_v0"): the barrier QP solved the way the reference's stack solves it --
 * rps hands `qp(H = 2I, f = -2 vec(uhat), A, b)` to cvxopt at reltol = feastol = 1e-2, maxiters 50 (utilities/controller.py:13-16,23
 * -> rps barrier_certificates), and cvxopt's coneqp returns an interior-point ITERATE that stops strictly
 * inside the feasible set.  This is the same iteration as barrier_qp_ipm above (restated coneqp for the linear cone: default starting
 * point, Mehrotra predictor-corrector, step 0.99 to the boundary, the same stopping rule), written as the explicit sequence of IEEE
 * BINARY64 operations the HIP kernels execute (csrc/ipm_qp.h) -- the float tier calls this very function on its binary32 xi and
 * uhat and rounds the result once, so kernels and oracle agree bit for bit.  Why binary64 inside a float spec: the KKT matrix
 * 2I + G' diag(z/s) G reaches condition numbers of 1e6 .. 1e8 near the boundary (z/s up to 1e7 under the 1e6 unsafe gain), and
 * cvxopt's stopping rule holds the dual residual against an ABSOLUTE 1e-2 while the multipliers are ~1e4: a binary32 transcription
 * misses the stop (residual noise 1.5e-2), runs on to maxiters and ends in NaN (measured, tests/test_ipm_spec.py).  MI355X
 * issues v_fma_f64 at the binary32 rate, so this costs registers, not issue slots.
 *   - rows keep rps' order (i < j, i outer); a row is never stored: G x and G' z are formed from e_c = xi_i - xi_j (exact in binary64);
 *   - K = 2I + G' diag(w) G is assembled in its 2 x 2 block form (diagonal blocks sum w e e' over the partners in row order, an
 *     off-diagonal block is -w e e' of its one pair) and factored K = L D L' without square roots; every solve is forward, scale,
 *     backward with fma chains in ascending index order;
 *   - ONE reciprocal per row per iteration (1 / s_c) and one per pivot; the step to the boundary max(-ds/s, -dz/z) is found by
 *     cross-multiplied comparisons (a/b > c/d  <=>  a d > c b, all denominators positive) in a fixed order (interleaved runs of
 *     rows + a balanced tree, see below) and divided once;
 *   - no division instruction: every reciprocal is ipm_rcp below (exponent-field seed + five Newton steps: deterministic, within
 *     two ulp; arguments are positive by construction -- s_c, z_c stay inside the cone, the pivots of 2I + PSD are >= 2);
 *   - no square root anywhere: the residual tests compare squared norms (|r| <= tol max(1, |r0|)  <=>  r.r <= tol^2 max(1, r0.r0));
 *   - sums run in ascending index order, left to right.
 * Like everything below the reference's own layers this is PARITY UNPINNED against real cvxopt; against barrier_qp_ipm it differs by
 * rounding only (tests/test_ipm_spec.py). */
#define ORC_IPM_MAXN 8
#define ORC_IPM_MAXV (2 * ORC_IPM_MAXN)
#define ORC_IPM_MAXC (ORC_IPM_MAXN * (ORC_IPM_MAXN - 1) / 2)
/* 1 / v for v > 0 without the divider: a seed from the exponent field (relative error < 12.5 %) and five Newton steps
 * r <- r + r (1 - v r), each two fmas: the error squares every step (1.6e-2, 2.4e-4, 6e-8, 3.6e-15, then rounding), so the result
 * is the reciprocal to within two units in the last place -- and the SAME BITS on every IEEE machine, which the hardware's
 * v_rcp_f64 is not.  A correctly rounded binary64 division costs ~30 dependent instructions on gfx950 (v_div_scale / v_rcp /
 * refinement / v_div_fmas through VCC, so independent divisions do not interleave); this costs 12 that do.  (ipm_spec_v0 only.) */
static inline double ipm_rcp(double v) {
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    b = 0x7FDE6238DA3C2118ull - b;
    double r;
    __builtin_memcpy(&r, &b, 8);
    for (int t = 0; t < 5; ++t) r = __builtin_fma(r, __builtin_fma(-v, r, 1.0), r);
    return r;
}
/* K (lower triangle, row-major n x n storage) <- 2I + sum_c w_c a_c a_c', a_c = -2 e_c at robot i, +2 e_c at robot j: with
 * W = 4 w the blocks are +-W e e' */
static void ipm_assemble(int N, const double *ex, const double *ey, const double *w4, double *K) {
    const int n = 2 * N;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c) K[r * n + c] = r == c ? 2.0 : 0.0;
    int c = 0;
    for (int i = 0; i < N - 1; ++i)
        for (int j = i + 1; j < N; ++j, ++c) {
            const double a = w4[c] * ex[c], b = w4[c] * ey[c];
            const double wxx = a * ex[c], wxy = a * ey[c], wyy = b * ey[c];
            K[(2 * i) * n + 2 * i] = K[(2 * i) * n + 2 * i] + wxx;
            K[(2 * i + 1) * n + 2 * i] = K[(2 * i + 1) * n + 2 * i] + wxy;
            K[(2 * i + 1) * n + 2 * i + 1] = K[(2 * i + 1) * n + 2 * i + 1] + wyy;
            K[(2 * j) * n + 2 * j] = K[(2 * j) * n + 2 * j] + wxx;
            K[(2 * j + 1) * n + 2 * j] = K[(2 * j + 1) * n + 2 * j] + wxy;
            K[(2 * j + 1) * n + 2 * j + 1] = K[(2 * j + 1) * n + 2 * j + 1] + wyy;
            K[(2 * j) * n + 2 * i] = -wxx;
            K[(2 * j) * n + 2 * i + 1] = -wxy;
            K[(2 * j + 1) * n + 2 * i] = -wxy;
            K[(2 * j + 1) * n + 2 * i + 1] = -wyy;
        }
}
/* in place: K <- unit lower L (below the diagonal), rd <- 1 / D */
static void ipm_ldl(int n, double *K, double *rd) {
    double v[ORC_IPM_MAXV];
    for (int j = 0; j < n; ++j) {
        double d = K[j * n + j];
        for (int k = 0; k < j; ++k) {
            v[k] = K[j * n + k] * K[k * n + k]; /* L_jk D_k */
            d = __builtin_fma(-K[j * n + k], v[k], d);
        }
        K[j * n + j] = d;
        rd[j] = ipm_rcp(d);
        for (int i = j + 1; i < n; ++i) {
            double t = K[i * n + j];
            for (int k = 0; k < j; ++k) t = __builtin_fma(-K[i * n + k], v[k], t);
            K[i * n + j] = t * rd[j];
        }
    }
}
static void ipm_solve(int n, const double *K, const double *rd, double *b) {
    for (int i = 1; i < n; ++i) {
        double t = b[i];
        for (int k = 0; k < i; ++k) t = __builtin_fma(-K[i * n + k], b[k], t);
        b[i] = t;
    }
    for (int i = 0; i < n; ++i) b[i] = b[i] * rd[i];
    for (int i = n - 2; i >= 0; --i) {
        double t = b[i];
        for (int k = i + 1; k < n; ++k) t = __builtin_fma(-K[k * n + i], b[k], t);
        b[i] = t;
    }
}
/* xi: the robots' single-integrator points; u: in, the control inputs ALREADY thresholded to the magnitude limit; out, the iterate.
 * Returns the iteration count (cvxopt's `iterations`). */
typedef struct { double abstol, reltol, feas2, r2, gain, ugain; int maxiters, has_unsafe; } ipm_consts;
/* as_float: the float tier's parameters are binary32 values (what the product's rg_scenario_params carries), widened exactly */
static ipm_consts ipm_make_consts(const orc_params *p, int as_float) {
#define IPM_P(v) (as_float ? (double)(float)(v) : (double)(v))
    ipm_consts k;
    k.abstol = IPM_P(p->ipm_abstol);
    k.reltol = IPM_P(p->ipm_reltol);
    k.feas2 = IPM_P(p->ipm_feastol) * IPM_P(p->ipm_feastol);
    k.r2 = IPM_P(p->safety_radius) * IPM_P(p->safety_radius);
    k.gain = IPM_P(p->barrier_gain);
    k.ugain = IPM_P(p->unsafe_barrier_gain);
    k.maxiters = p->ipm_maxiters;
    k.has_unsafe = p->barrier_has_unsafe_gain;
#undef IPM_P
    return k;
}
static int barrier_qp_ipm_spec(const ipm_consts *kc, int N, const double *xix, const double *xiy, double *ux, double *uy) {
    if (N < 2 || N > ORC_IPM_MAXN) return N < 2 ? 0 : -1; /* no rows: the unconstrained minimiser is the input itself */
    const int n = 2 * N, m = N * (N - 1) / 2;
    const double ABSTOL = kc->abstol, RELTOL = kc->reltol, FEAS2 = kc->feas2, STEP = 0.99;
    const int MAXITERS = kc->maxiters;
    const double r2 = kc->r2;
    double ex[ORC_IPM_MAXC], ey[ORC_IPM_MAXC], h[ORC_IPM_MAXC], s[ORC_IPM_MAXC], z[ORC_IPM_MAXC], w4[ORC_IPM_MAXC];
    double rz[ORC_IPM_MAXC], rs[ORC_IPM_MAXC], ds[ORC_IPM_MAXC], dz[ORC_IPM_MAXC], dsdza[ORC_IPM_MAXC];
    double q[ORC_IPM_MAXV], x[ORC_IPM_MAXV], rx[ORC_IPM_MAXV], dx[ORC_IPM_MAXV], rd[ORC_IPM_MAXV], K[ORC_IPM_MAXV * ORC_IPM_MAXV];
    double nh = 0.0;
    {
        int c = 0;
        for (int i = 0; i < N - 1; ++i)
            for (int j = i + 1; j < N; ++j, ++c) {
                ex[c] = xix[i] - xix[j];
                ey[c] = xiy[i] - xiy[j];
                const double hh = __builtin_fma(ex[c], ex[c], ey[c] * ey[c]) - r2;
                const double gain = (hh >= 0.0 || !kc->has_unsafe) ? kc->gain : kc->ugain;
                h[c] = gain * ((hh * hh) * hh);
                nh = __builtin_fma(h[c], h[c], nh);
                w4[c] = 4.0;
            }
    }
    double nq = 0.0;
    for (int a = 0; a < N; ++a) {
        q[2 * a] = -2.0 * ux[a];
        q[2 * a + 1] = -2.0 * uy[a];
        nq = __builtin_fma(q[2 * a], q[2 * a], nq);
        nq = __builtin_fma(q[2 * a + 1], q[2 * a + 1], nq);
    }
    const double resx0sq = FEAS2 * (nq > 1.0 ? nq : 1.0), resz0sq = FEAS2 * (nh > 1.0 ? nh : 1.0);
    /* default starting point: (2I + G'G) x = -q + G'h;  z = G x - h;  s = -z;  both shifted into the cone if they are not inside */
    ipm_assemble(N, ex, ey, w4, K);
    ipm_ldl(n, K, rd);
    for (int k = 0; k < n; ++k) x[k] = -q[k];
    {
        int c = 0;
        for (int i = 0; i < N - 1; ++i)
            for (int j = i + 1; j < N; ++j, ++c) {
                const double t = 2.0 * h[c];
                x[2 * i] = __builtin_fma(-t, ex[c], x[2 * i]);
                x[2 * i + 1] = __builtin_fma(-t, ey[c], x[2 * i + 1]);
                x[2 * j] = __builtin_fma(t, ex[c], x[2 * j]);
                x[2 * j + 1] = __builtin_fma(t, ey[c], x[2 * j + 1]);
            }
    }
    ipm_solve(n, K, rd, x);
    double gap;
    {
        double ns = 0.0, tz = -1e300, ts = -1e300;
        int c = 0;
        for (int i = 0; i < N - 1; ++i)
            for (int j = i + 1; j < N; ++j, ++c) {
                const double gx_ = 2.0 * __builtin_fma(ex[c], x[2 * j] - x[2 * i], ey[c] * (x[2 * j + 1] - x[2 * i + 1]));
                z[c] = gx_ - h[c];
                s[c] = -z[c];
                ns = __builtin_fma(z[c], z[c], ns);
                ts = z[c] > ts ? z[c] : ts;   /* max(-s) */
                tz = s[c] > tz ? s[c] : tz;   /* max(-z) */
            }
        /* t >= -1e-8 max(|s|, 1), without the root: t >= 0, or t^2 <= 1e-16 max(s.s, 1) */
        const double lim2 = 1e-16 * (ns > 1.0 ? ns : 1.0);
        const int shift_s = ts >= 0.0 || ts * ts <= lim2, shift_z = tz >= 0.0 || tz * tz <= lim2;
        const double as = 1.0 + ts, az = 1.0 + tz;
        gap = 0.0;
        for (c = 0; c < m; ++c) {
            if (shift_s) s[c] = s[c] + as;
            if (shift_z) z[c] = z[c] + az;
            gap = __builtin_fma(s[c], z[c], gap);
        }
    }
    int iters = 0;
    for (;; ++iters) {
        /* residuals: rx = q + 2x + G'z, rz = s - h + G x; costs */
        double f0 = 0.0, nrx = 0.0, nrz = 0.0, zrz = 0.0;
        for (int k = 0; k < n; ++k) {
            rx[k] = __builtin_fma(2.0, x[k], q[k]);
            f0 = __builtin_fma(x[k], rx[k] + q[k], f0); /* x'(Px + q) + x'q, halved below */
        }
        f0 = 0.5 * f0;
        {
            int c = 0;
            for (int i = 0; i < N - 1; ++i)
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
        for (int k = 0; k < n; ++k) nrx = __builtin_fma(rx[k], rx[k], nrx);
        const double pcost = f0, dcost = (f0 + zrz) - gap;
        /* relgap <= reltol without a division: gap <= reltol * (-pcost) or gap <= reltol * dcost */
        const int rel_ok = pcost < 0.0 ? gap <= RELTOL * -pcost : dcost > 0.0 ? gap <= RELTOL * dcost : 0;
#ifdef ORC_IPM_TRACE
        fprintf(stderr, "spec it %2d gap %.6e pcost %.6e dcost %.6e resx %.3e resz %.3e x0 %.9e\n", iters, gap, pcost, dcost, sqrt(nrx), sqrt(nrz), x[0]);
#endif
        if ((nrz <= resz0sq && nrx <= resx0sq && (gap <= ABSTOL || rel_ok)) || iters == MAXITERS) break;
        /* scaling and the KKT matrix of this iteration */
        for (int c = 0; c < m; ++c) {
            rs[c] = ipm_rcp(s[c]);
            w4[c] = 4.0 * (z[c] * rs[c]);
        }
        ipm_assemble(N, ex, ey, w4, K);
        ipm_ldl(n, K, rd);
        const double mu = gap * (1.0 / (double)m);
        double sigmamu = 0.0, step = 1.0;
        for (int pass = 0; pass < 2; ++pass) {
            /* rc = -s o z + sigma mu [- dsa o dza];  dx = K^-1 (-rx - G'[(rc + z o rz) / s]) */
            for (int k = 0; k < n; ++k) dx[k] = -rx[k];
            {
                int c = 0;
                for (int i = 0; i < N - 1; ++i)
                    for (int j = i + 1; j < N; ++j, ++c) {
                        const double rc = pass ? __builtin_fma(-s[c], z[c], sigmamu) - dsdza[c] : -(s[c] * z[c]);
                        const double t = 2.0 * (__builtin_fma(z[c], rz[c], rc) * rs[c]);
                        dx[2 * i] = __builtin_fma(t, ex[c], dx[2 * i]);        /* - G'(.): the row holds -2e at i, +2e at j */
                        dx[2 * i + 1] = __builtin_fma(t, ey[c], dx[2 * i + 1]);
                        dx[2 * j] = __builtin_fma(-t, ex[c], dx[2 * j]);
                        dx[2 * j + 1] = __builtin_fma(-t, ey[c], dx[2 * j + 1]);
                    }
            }
            ipm_solve(n, K, rd, dx);
            /* ds = -rz - G dx;  dz = (rc - z o ds) / s;  step to the boundary t = max(0, -ds/s, -dz/z) as a fraction tn / td.
             * The maximum is taken over LANES interleaved runs of rows (run l: rows l, l + LANES, ...; each run folds its rows'
             * two candidates in order, starting from 0 / 1) whose results meet in a balanced tree (0 with 1, 2 with 3, ...; the
             * lower run is kept unless the higher one is strictly larger): LANES = 1 up to four robots, 8 from five on -- the
             * kernel's lane groups fold their own rows and meet by lane permutes, a chain of 2 + 3 comparisons instead of 2m. */
            double dsdz = 0.0, tn, td;
            {
                int c = 0;
                for (int i = 0; i < N - 1; ++i)
                    for (int j = i + 1; j < N; ++j, ++c) {
                        const double gdx = 2.0 * __builtin_fma(ex[c], dx[2 * j] - dx[2 * i], ey[c] * (dx[2 * j + 1] - dx[2 * i + 1]));
                        const double rc = pass ? __builtin_fma(-s[c], z[c], sigmamu) - dsdza[c] : -(s[c] * z[c]);
                        ds[c] = -rz[c] - gdx;
                        dz[c] = __builtin_fma(-z[c], ds[c], rc) * rs[c];
                        dsdz = __builtin_fma(ds[c], dz[c], dsdz);
                    }
                const int LANES = N <= 4 ? 1 : 8;
                double pn[8], pd[8];
                for (int l = 0; l < LANES; ++l) {
                    pn[l] = 0.0;
                    pd[l] = 1.0;
                    for (c = l; c < m; c += LANES) {
                        if (-ds[c] * pd[l] > pn[l] * s[c]) { pn[l] = -ds[c]; pd[l] = s[c]; }
                        if (-dz[c] * pd[l] > pn[l] * z[c]) { pn[l] = -dz[c]; pd[l] = z[c]; }
                    }
                }
                for (int w = 1; w < LANES; w *= 2)
                    for (int l = 0; l < LANES; l += 2 * w)
                        if (pn[l + w] * pd[l] > pn[l] * pd[l + w]) { pn[l] = pn[l + w]; pd[l] = pd[l + w]; }
                tn = pn[0];
                td = pd[0];
            }
            if (pass == 0) {
                step = tn > td ? td * ipm_rcp(tn) : 1.0;         /* min(1, 1 / t);  t == 0 -> 1 */
                double sg = __builtin_fma(dsdz * ipm_rcp(gap), step * step, 1.0 - step);
                sg = sg < 0.0 ? 0.0 : sg > 1.0 ? 1.0 : sg;
                sigmamu = ((sg * sg) * sg) * mu;
                for (int c = 0; c < m; ++c) dsdza[c] = ds[c] * dz[c];
            } else {
                step = STEP * td < tn ? (STEP * td) * ipm_rcp(tn) : 1.0;   /* min(1, 0.99 / t) */
            }
        }
        for (int k = 0; k < n; ++k) x[k] = __builtin_fma(step, dx[k], x[k]);
        gap = 0.0;
        for (int c = 0; c < m; ++c) {
            s[c] = __builtin_fma(step, ds[c], s[c]);
            z[c] = __builtin_fma(step, dz[c], z[c]);
            gap = __builtin_fma(s[c], z[c], gap);
        }
    }
    for (int a = 0; a < N; ++a) {
        ux[a] = x[2 * a];
        uy[a] = x[2 * a + 1];
    }
    return iters;
}
#endif

/* the float tier's entry (and the float64 tier's qp_mode 2): threshold in the tier's own arithmetic, solve in binary64, round once */
static int FN(barrier_qp_ipm_spec_call)(const orc_params *p, int N, const REAL *xix, const REAL *xiy, REAL *ux, REAL *uy) {
    double dxi[ORC_MAXN], dyi[ORC_MAXN], dux[ORC_MAXN], duy[ORC_MAXN];
    for (int a = 0; a < N; ++a) { /* "Threshold control inputs before QP", decided on squares like the exact mode */
        REAL n2u = ux[a] * ux[a] + uy[a] * uy[a];
        if (n2u > R(p->barrier_magnitude_limit) * R(p->barrier_magnitude_limit)) {
            REAL sc = R(p->barrier_magnitude_limit) / SQRT(n2u);
            ux[a] = ux[a] * sc;
            uy[a] = uy[a] * sc;
        }
        dxi[a] = (double)xix[a];
        dyi[a] = (double)xiy[a];
        dux[a] = (double)ux[a];
        duy[a] = (double)uy[a];
    }
    const ipm_consts kc = ipm_make_consts(p, ORC_IS_F32);
    int it = barrier_qp_ipm_spec(&kc, N, dxi, dyi, dux, duy);
    for (int a = 0; a < N; ++a) {
        ux[a] = (REAL)dux[a];
        uy[a] = (REAL)duy[a];
    }
    return it;
}

/* a3 = a4 . a5 . a6 . a7, then a8 (utilities/controller.py:20-24, roboEnv.py:64-65) */
static int FN(controller)(const orc_params *p, int N, const REAL *x, const REAL *y, const REAL *cs, const REAL *ss,
                          const REAL *gx, const REAL *gy, REAL *v, REAL *w) {
    REAL xix[ORC_MAXN], xiy[ORC_MAXN], ux[ORC_MAXN], uy[ORC_MAXN];
    REAL pd = R(p->projection_distance);
    for (int a = 0; a < N; ++a) {
        xix[a] = x[a] + pd * cs[a]; /* a4 uni_to_si_states */
        xiy[a] = y[a] + pd * ss[a];
        REAL dx = gx[a] - xix[a], dy = gy[a] - xiy[a]; /* a5 si_position_controller, gain 1 */
        REAL nrm = SQRT(dx * dx + dy * dy);
        if (nrm > R(p->position_velocity_limit)) {
            REAL sc = R(p->position_velocity_limit) / nrm;
            dx = dx * sc;
            dy = dy * sc;
        }
        ux[a] = dx;
        uy[a] = dy;
    }
    /* a6.  qp_mode 0: the exact projection; 1: the restated cvxopt iterate -- in its own operation order in the float64 tier, as
     * ipm_spec_v0 in the float tier (= the kernels); 2: ipm_spec_v0 in whichever precision (float64: the study twin of the spec) */
#if !ORC_IS_F32
    int sweeps = p->qp_mode == 1 ? FN(barrier_qp_ipm)(p, N, xix, xiy, ux, uy)
               : p->qp_mode == 2 ? FN(barrier_qp_ipm_spec_call)(p, N, xix, xiy, ux, uy) : FN(barrier_qp)(p, N, xix, xiy, ux, uy);
#else
    int sweeps = p->qp_mode != 0 ? FN(barrier_qp_ipm_spec_call)(p, N, xix, xiy, ux, uy) : FN(barrier_qp)(p, N, xix, xiy, ux, uy);
#endif
    REAL inv_pd = R(1) / pd;
    REAL wlim = R(p->angular_velocity_limit);
    REAL vmax = R(p->max_linear_velocity);
    REAL wmax = R(2) * (R(p->wheel_radius) / R(p->robot_diameter)) * (vmax / R(p->wheel_radius));
    for (int a = 0; a < N; ++a) { /* a7 si_to_uni_dyn, a8 set_velocities */
        REAL vv = cs[a] * ux[a] + ss[a] * uy[a];
        REAL ww = inv_pd * (-ss[a] * ux[a] + cs[a] * uy[a]);
        if (ww > wlim) ww = wlim;
        if (ww < -wlim) ww = -wlim;
        if (vv > vmax) vv = vmax;
        if (vv < -vmax) vv = -vmax;
        if (ww > wmax) ww = wmax;
        if (ww < -wmax) ww = -wmax;
        v[a] = vv;
        w[a] = ww;
    }
    return sweeps;
}

/* a10 _validate: returns bit0 collision, bit1 boundary */
static int FN(validate)(const orc_params *p, int N, const REAL *x, const REAL *y, const REAL *cs, const REAL *ss) {
    int code = 0;
    REAL xmin = R(p->bound_x0), ymin = R(p->bound_y0);
    REAL xmax = xmin + R(p->bound_w), ymax = ymin + R(p->bound_h);
    for (int a = 0; a < N; ++a)
        if (x[a] < xmin || x[a] > xmax || y[a] < ymin || y[a] > ymax) code |= 2;
    REAL fx[ORC_MAXN], fy[ORC_MAXN];
    REAL lim;
    if (p->collision_variant == 1) {
        for (int a = 0; a < N; ++a) {
#if ORC_IS_F32
            fx[a] = __builtin_fmaf(R(p->collision_offset), cs[a], x[a]);
            fy[a] = __builtin_fmaf(R(p->collision_offset), ss[a], y[a]);
#else
            fx[a] = x[a] + R(p->collision_offset) * cs[a];
            fy[a] = y[a] + R(p->collision_offset) * ss[a];
#endif
        }
        lim = R(p->collision_diameter);
    } else {
        for (int a = 0; a < N; ++a) {
            fx[a] = x[a];
            fy[a] = y[a];
        }
        lim = R(p->robot_diameter);
    }
    for (int j = 0; j < N - 1; ++j)
        for (int k = j + 1; k < N; ++k) {
            REAL dx = fx[j] - fx[k], dy = fy[j] - fy[k];
#if ORC_IS_F32
            if (dx * dx + dy * dy <= lim * lim) code |= 1; /* float spec: squared form, no sqrt per pair */
#else
            if (SQRT(dx * dx + dy * dy) <= lim) code |= 1;
#endif
        }
    return code;
}

/* neighbours of agent a: K nearest, ascending distance, ties -> lower index (the canonical
 * order; misc.py:20-25 leaves it to np.argpartition).  K >= N-1: all others in index order. */
static int FN(neighbours)(int N, int K, int a, const REAL *x, const REAL *y, int *out) {
    if (K >= N - 1) {
        int n = 0;
        for (int j = 0; j < N; ++j)
            if (j != a) out[n++] = j;
        return n;
    }
    REAL d[ORC_MAXN];
    for (int j = 0; j < N; ++j) {
        REAL dx = x[j] - x[a], dy = y[j] - y[a];
        d[j] = DIST(dx * dx + dy * dy); /* float spec: squared distances (same order, no sqrt) */
    }
    for (int j = 0; j < N; ++j) {
        if (j == a) continue;
        int rank = 0;
        for (int k = 0; k < N; ++k) {
            if (k == a || k == j) continue;
            if (d[k] < d[j] || (d[k] == d[j] && k < j)) ++rank;
        }
        if (rank < K) out[rank] = j;
    }
    return K;
}

static void FN(step_env)(const orc_params *p, int e, const FN(orc_state) * st, const int32_t *actions,
                         const FN(orc_out) * out) {
    const int N = p->n_agents;
    REAL *X = st->poses + (size_t)e * 3 * N, *Y = X + N, *TH = Y + N;
    REAL x[ORC_MAXN], y[ORC_MAXN], th[ORC_MAXN], gx[ORC_MAXN], gy[ORC_MAXN], px[ORC_MAXN], py[ORC_MAXN];
    REAL v[ORC_MAXN], w[ORC_MAXN], dist[ORC_MAXN];
    const int32_t *act = actions + (size_t)e * N;
    REAL L = R(p->left), Rt = R(p->right), Up = R(p->up), Dn = R(p->down);
    for (int a = 0; a < N; ++a) {
        x[a] = X[a];
        y[a] = Y[a];
        th[a] = TH[a];
        v[a] = R(0);
        w[a] = R(0);
        dist[a] = R(0);
        px[a] = x[a];
        py[a] = y[a];
        /* a1 generate_goal (PCP agent.py:48-76, warehouse.py:19-45, MaterialTransport.py:19-46) */
        int mv = (p->scenario == ORC_SCN_MT) ? act[a] / 4 : act[a];
        REAL sd = R(p->agent_step[a]);
        if (p->scenario == ORC_SCN_ARCTIC) { /* ArcticTransport/agent.py:89-113: drones 0,1; ice 2; water 3 */
            int pix = st->pixel_type[(size_t)e * N + a];
            if (a < 2) sd = R(p->arctic_fast_step);
            else if (a == 3) sd = pix == 1 ? R(p->arctic_slow_step) : pix == 2 ? R(p->arctic_fast_step) : R(p->arctic_normal_step);
            else sd = pix == 1 ? R(p->arctic_fast_step) : pix == 2 ? R(p->arctic_slow_step) : R(p->arctic_normal_step);
        }
        REAL tx = x[a], ty = y[a];
        if (mv == 0) {
            REAL t = tx - sd;
            tx = t > L ? t : L;
            ty = FN(clampv)(ty, Up, Dn);
        } else if (mv == 1) {
            REAL t = tx + sd;
            tx = t < Rt ? t : Rt;
            ty = FN(clampv)(ty, Up, Dn);
        } else if (mv == 2) {
            tx = FN(clampv)(tx, L, Rt);
            REAL t = ty - sd;
            ty = t > Up ? t : Up;
        } else if (mv == 3) {
            tx = FN(clampv)(tx, L, Rt);
            REAL t = ty + sd;
            ty = t < Dn ? t : Dn;
        } else {
            tx = FN(clampv)(tx, L, Rt);
            ty = FN(clampv)(ty, Up, Dn);
        }
        gx[a] = tx;
        gy[a] = ty;
    }
    /* a2 roboEnv.step (utilities/roboEnv.py:38-96) */
    (void)px;
    (void)py;
    (void)dist;
    int viol = 0, max_sweeps = 0;
    REAL dt = R(p->time_step);
    REAL cs[ORC_MAXN], ss[ORC_MAXN];
#if ORC_IS_F32
    /* float spec (sim_spec_v0), algebraically the same step as the float64 sequence below, arranged
     * per CONTROLLER PERIOD (the <= 15 sub-steps during which v and w are held, roboEnv.py:63):
     *  - the length of a sub-step is |dt*v| (= |P_k+1 - P_k| of the Euler update): no sqrt, and n
     *    equal sub-steps add n*|dt*v| in one fma.  dist_travelled lags one sub-step
     *    (roboEnv.py:55-59): dist = carry_in + sum of all sub-step lengths - the last one, which
     *    becomes carry_out; on a violation the last one is included (roboEnv.py:93);
     *  - theta advances n*dt*w in one fma and is wrapped once per period;
     *  - within a period sin/cos of the heading advance by the rotation (cos, sin)(dt*w) instead
     *    of being re-evaluated; the collision offset point uses explicit fma;
     *  - x, y are kept as a period base (bx, by) plus the displacement since the period began
     *    (ox, oy: |.| <= 15 * 0.0066 m, so its roundings are ~16x finer than those of a coordinate
     *    near 1 m): a sub-step adds dt*v*(cos, sin) to the displacement with one fma and the
     *    position of sub-step j is the single rounding bx + ox -- rounding errors of the 29..74
     *    Euler updates do not pile up in x, y (they did, and a reversing unicycle amplifies them:
     *    measured 7x smaller |x - x_f64| per step, tests/golden/PARITY_REPORT.json).  At a period
     *    end the base moves to bx + ox and the exact remainder of that addition (TwoSum) seeds the
     *    next period's displacement. */
    REAL acc[ORC_MAXN], dtv[ORC_MAXN], dtw[ORC_MAXN], cd[ORC_MAXN], sd[ORC_MAXN], last[ORC_MAXN];
    REAL bx[ORC_MAXN], by[ORC_MAXN], ox[ORC_MAXN], oy[ORC_MAXN];
    for (int a = 0; a < N; ++a) {
        acc[a] = st->carry[(size_t)e * N + a];
        last[a] = R(0);
        bx[a] = x[a];
        by[a] = y[a];
        ox[a] = R(0);
        oy[a] = R(0);
    }
    for (int it0 = 0; it0 < p->update_frequency && !viol; it0 += p->controller_period) {
        int n = p->update_frequency - it0;
        if (n > p->controller_period) n = p->controller_period;
        for (int a = 0; a < N; ++a) SINCOS(th[a], &ss[a], &cs[a]);
        int sw = FN(controller)(p, N, x, y, cs, ss, gx, gy, v, w);
        if (sw > max_sweeps) max_sweeps = sw;
        for (int a = 0; a < N; ++a) {
            dtv[a] = dt * v[a];
            dtw[a] = dt * w[a];
            SINCOS_STEP(dtw[a], &sd[a], &cd[a]);
        }
        int n_exec = n;
        for (int j = 0; j < n; ++j) {
            int code = FN(validate)(p, N, x, y, cs, ss);
            for (int a = 0; a < N; ++a) {
                ox[a] = __builtin_fmaf(cs[a], dtv[a], ox[a]);
                oy[a] = __builtin_fmaf(ss[a], dtv[a], oy[a]);
                x[a] = bx[a] + ox[a];
                y[a] = by[a] + oy[a];
                REAL cn = __builtin_fmaf(cs[a], cd[a], -(ss[a] * sd[a]));
                REAL sn = __builtin_fmaf(ss[a], cd[a], cs[a] * sd[a]);
                cs[a] = cn;
                ss[a] = sn;
            }
            if (p->penalize_violations && code) {
                viol = code;
                n_exec = j + 1;
                break;
            }
        }
        for (int a = 0; a < N; ++a) {
            last[a] = __builtin_fabsf(dtv[a]);
            th[a] = WRAP(__builtin_fmaf((REAL)n_exec, dtw[a], th[a]));
            acc[a] = __builtin_fmaf((REAL)n_exec, last[a], acc[a]);
            /* base <- base + displacement (= x, y as last formed); displacement <- the exact remainder */
            REAL tx = x[a] - bx[a], ty = y[a] - by[a];
            ox[a] = (bx[a] - (x[a] - tx)) + (ox[a] - tx);
            oy[a] = (by[a] - (y[a] - ty)) + (oy[a] - ty);
            bx[a] = x[a];
            by[a] = y[a];
        }
    }
    for (int a = 0; a < N; ++a) {
        X[a] = x[a];
        Y[a] = y[a];
        TH[a] = th[a];
        st->carry[(size_t)e * N + a] = last[a];
        out->dist[(size_t)e * N + a] = viol ? acc[a] : acc[a] - last[a];
    }
#else
    for (int it = 0; it < p->update_frequency; ++it) {
        for (int a = 0; a < N; ++a) {
            if (it == 0) {
                dist[a] = dist[a] + st->carry[(size_t)e * N + a];
            } else {
                REAL dx = x[a] - px[a], dy = y[a] - py[a];
                dist[a] = dist[a] + SQRT(dx * dx + dy * dy);
            }
            px[a] = x[a];
            py[a] = y[a];
        }
        for (int a = 0; a < N; ++a) SINCOS(th[a], &ss[a], &cs[a]);
        if (it % p->controller_period == 0) {
            int s = FN(controller)(p, N, x, y, cs, ss, gx, gy, v, w);
            if (s > max_sweeps) max_sweeps = s;
        }
        int code = FN(validate)(p, N, x, y, cs, ss); /* rps step(): validate first ... */
        for (int a = 0; a < N; ++a) {              /* ... then Euler + wrap (Appendix A.4) */
            x[a] = x[a] + dt * cs[a] * v[a];
            y[a] = y[a] + dt * ss[a] * v[a];
            REAL t = th[a] + dt * w[a];
            th[a] = WRAP(t);
        }
        if (p->penalize_violations && code) { /* roboEnv.py:82-94 */
            viol = code;
            for (int a = 0; a < N; ++a) {
                REAL dx = x[a] - px[a], dy = y[a] - py[a];
                dist[a] = dist[a] + SQRT(dx * dx + dy * dy);
            }
            break;
        }
    }
    for (int a = 0; a < N; ++a) {
        X[a] = x[a];
        Y[a] = y[a];
        TH[a] = th[a];
        REAL dx = x[a] - px[a], dy = y[a] - py[a];
        st->carry[(size_t)e * N + a] = SQRT(dx * dx + dy * dy);
        out->dist[(size_t)e * N + a] = dist[a];
    }
#endif
    out->viol[e] = (uint8_t)viol;
    if (out->qp_sweeps) out->qp_sweeps[e] = max_sweeps;
    int steps = st->steps[e] + 1;
    st->steps[e] = steps;
    const int D = p->obs_dim;
    REAL *obs = out->obs + (size_t)e * N * D;
    REAL *rew = out->reward + (size_t)e * N;
    int done = 0, remaining = -1;

    if (p->scenario == ORC_SCN_PCP) {
        const int P = p->num_prey;
        REAL *pl = st->prey_loc + (size_t)e * P * 2;
        uint8_t *sensed = st->prey_sensed + (size_t)e * P, *captured = st->prey_captured + (size_t)e * P;
        int unseen0 = 0, left0 = 0;
        for (int i = 0; i < P; ++i) {
            unseen0 += !sensed[i];
            left0 += !captured[i];
        }
        REAL dpa[ORC_MAXP][ORC_MAXN];
        for (int i = 0; i < P; ++i)
            for (int a = 0; a < N; ++a) {
                REAL dx = x[a] - pl[2 * i], dy = y[a] - pl[2 * i + 1];
                dpa[i][a] = DIST(dx * dx + dy * dy);
            }
        /* a11 _update_tracking_and_locations (PredatorCapturePrey.py:72-95) */
        for (int i = 0; i < P; ++i) {
            if (captured[i]) continue;
            if (!sensed[i])
                for (int a = 0; a < N; ++a)
                    if (dpa[i][a] <= RADIUS(R(p->sensing_radius[a]))) {
                        sensed[i] = 1;
                        break;
                    }
            if (sensed[i])
                for (int a = 0; a < N; ++a)
                    if (act[a] == 4 && dpa[i][a] <= RADIUS(R(p->capture_radius[a]))) {
                        captured[i] = 1;
                        break;
                    }
        }
        int unseen1 = 0, left1 = 0; /* a12 */
        for (int i = 0; i < P; ++i) {
            unseen1 += !sensed[i];
            left1 += !captured[i];
        }
        /* a13 observations (agent.py:19-46, PredatorCapturePrey.py:178-207) */
        const int od = p->capability_aware ? 6 : 4;
        REAL own[ORC_MAXN][6];
        for (int a = 0; a < N; ++a) {
            REAL closest = R(-1), qx = R(-5), qy = R(-5);
            for (int i = 0; i < P; ++i) {
                if (captured[i]) continue;
                REAL d = dpa[i][a];
                if (d <= RADIUS(R(p->sensing_radius[a])) && (d < closest || closest == R(-1))) {
                    qx = pl[2 * i];
                    qy = pl[2 * i + 1];
                    closest = d;
                }
            }
            own[a][0] = x[a];
            own[a][1] = y[a];
            own[a][2] = qx;
            own[a][3] = qy;
            own[a][4] = R(p->sensing_radius[a]);
            own[a][5] = R(p->capture_radius[a]);
        }
        for (int a = 0; a < N; ++a) {
            int nb[ORC_MAXN];
            int nn = FN(neighbours)(N, p->num_neighbors, a, x, y, nb);
            REAL *o = obs + (size_t)a * D;
            for (int c = 0; c < od; ++c) o[c] = own[a][c];
            for (int m = 0; m < nn; ++m)
                for (int c = 0; c < od; ++c) o[(m + 1) * od + c] = own[nb[m]][c];
        }
        /* a14 reward / termination (PredatorCapturePrey.py:155-176, 209-216) */
        REAL r;
        if (viol) {
            r = R(p->violation_reward);
            done = 1;
        } else {
            r = R(0);
            r = r + (REAL)(unseen0 - unseen1) * R(p->sense_reward);
            r = r + (REAL)(left0 - left1) * R(p->capture_reward);
            r = r + R(p->time_penalty);
            if (steps > p->max_episode_steps || left1 == 0) {
                done = 1;
                remaining = left1;
            }
        }
        for (int a = 0; a < N; ++a) rew[a] = r;
    } else if (p->scenario == ORC_SCN_WAREHOUSE) {
        /* a15 (warehouse.py:124-143 obs BEFORE the reward mutates `loaded`, :145-178, :102-122) */
        uint8_t *loaded = st->loaded + (size_t)e * N;
        REAL own[ORC_MAXN][3];
        for (int a = 0; a < N; ++a) {
            own[a][0] = x[a];
            own[a][1] = y[a];
            own[a][2] = loaded[a] ? R(1) : R(0);
        }
        for (int a = 0; a < N; ++a) {
            int nb[ORC_MAXN];
            int nn = FN(neighbours)(N, p->num_neighbors, a, x, y, nb);
            REAL *o = obs + (size_t)a * D;
            for (int c = 0; c < 3; ++c) o[c] = own[a][c];
            for (int m = 0; m < nn; ++m)
                for (int c = 0; c < 3; ++c) o[(m + 1) * 3 + c] = own[nb[m]][c];
        }
        if (viol) {
            for (int a = 0; a < N; ++a) rew[a] = R(p->violation_reward);
            done = 1;
        } else {
            REAL gw_ = R(p->goal_width);
            for (int a = 0; a < N; ++a) {
                int green = (a % 2 == 0);
                REAL r = R(0);
                if (loaded[a]) {
                    if (x[a] < R(-1.5) + gw_) {
                        if ((green && y[a] > R(0)) || (!green && y[a] <= R(0))) {
                            r = R(p->unload_reward);
                            loaded[a] = 0;
                        }
                    }
                } else {
                    if (x[a] > R(1.5) - gw_) {
                        if ((!green && y[a] > R(0)) || (green && y[a] <= R(0))) {
                            r = R(p->load_reward);
                            loaded[a] = 1;
                        }
                    }
                }
                rew[a] = r;
            }
            done = steps > p->max_episode_steps;
        }
    } else if (p->scenario == ORC_SCN_SIMPLE) {
        /* Simple (scenarios/Simple/simple.py:155-225): obs = own (x,y), the others in index order,
         * the goal; dense per-agent reward -|p - goal|^2 * reward_scaler */
        REAL *gl = st->prey_loc + (size_t)e * 2;
        for (int a = 0; a < N; ++a) {
            REAL *o = obs + (size_t)a * D;
            int n = 0;
            o[n++] = x[a];
            o[n++] = y[a];
            for (int j = 0; j < N; ++j)
                if (j != a) {
                    o[n++] = x[j];
                    o[n++] = y[j];
                }
            o[n++] = gl[0];
            o[n++] = gl[1];
        }
        if (viol) {
            for (int a = 0; a < N; ++a) rew[a] = R(p->violation_reward);
            done = 1;
        } else {
            for (int a = 0; a < N; ++a) {
                REAL dx = x[a] - gl[0], dy = y[a] - gl[1];
                REAL r = -(dx * dx + dy * dy);
                rew[a] = r * R(p->reward_scaler);
            }
            done = steps > p->max_episode_steps;
        }
    } else if (p->scenario == ORC_SCN_ARCTIC) {
        /* ArcticTransport (ArcticTransport.py:84-143, agent.py:14-87) */
        const uint8_t *grid = st->grid + (size_t)e * 96;
        uint8_t *pix = st->pixel_type + (size_t)e * N, *reached = st->reached_goal + (size_t)e * N;
        int gc = st->goal_col[e];
        int row[4], col[4];
        for (int a = 0; a < 4; ++a) { /* get_cell_from_pose: int() truncates toward zero */
            int r_ = -(int)((y[a] - R(1)) / R(0.25)), c_ = (int)((x[a] + R(1.5)) / R(0.25));
            row[a] = r_ < 0 ? 0 : r_ > 7 ? 7 : r_;
            col[a] = c_ < 0 ? 0 : c_ > 11 ? 11 : c_;
        }
        REAL goalx = (REAL)gc * R(0.25) - R(1.5), goaly = R(-1) * R(0.25) + R(0.75);
        static const int others[4][3] = {{1, 2, 3}, {0, 2, 3}, {3, 0, 1}, {2, 0, 1}};
        for (int a = 0; a < 4; ++a) {
            REAL *o = obs + (size_t)a * D;
            int n = 0;
            pix[a] = grid[row[a] * 12 + col[a]];
            if (pix[a] == 3) reached[a] = 1;
            o[n++] = x[a];
            o[n++] = y[a];
            o[n++] = (REAL)pix[a];
            for (int m = 0; m < 3; ++m) {
                int j = others[a][m];
                o[n++] = x[j];
                o[n++] = y[j];
                o[n++] = (REAL)grid[row[j] * 12 + col[j]];
            }
            o[n++] = goalx;
            o[n++] = goaly;
            for (int i = 0; i < 2; ++i) { /* the 8 cells around each drone, edges clamped */
                int left = col[i] > 0 ? col[i] - 1 : col[i], right = col[i] < 11 ? col[i] + 1 : col[i];
                int up = row[i] > 0 ? row[i] - 1 : row[i], down = row[i] < 7 ? row[i] + 1 : row[i];
                o[n++] = (REAL)grid[up * 12 + left];
                o[n++] = (REAL)grid[row[i] * 12 + left];
                o[n++] = (REAL)grid[down * 12 + left];
                o[n++] = (REAL)grid[up * 12 + col[i]];
                o[n++] = (REAL)grid[down * 12 + col[i]];
                o[n++] = (REAL)grid[up * 12 + right];
                o[n++] = (REAL)grid[row[i] * 12 + right];
                o[n++] = (REAL)grid[down * 12 + right];
            }
        }
        REAL r;
        if (viol) {
            r = R(p->violation_reward);
            done = 1;
        } else {
            r = R(0);
            for (int a = 2; a < 4; ++a) { /* only the non-drones count */
                if (!reached[a]) r = r + R(p->not_reached_penalty);
                if (pix[a] != 3) {
                    REAL dx = x[a] - goalx, dy = y[a] - goaly;
#if ORC_IS_F32
                    r = r + R(p->dist_multiplier) * (dx * dx + dy * dy); /* float spec: dist^2 without the sqrt */
#else
                    REAL dd = SQRT(dx * dx + dy * dy);
                    r = r + R(p->dist_multiplier) * (dd * dd);
#endif
                }
            }
            done = steps > p->max_episode_steps;
            if (!done) done = reached[2] && reached[3];
        }
        for (int a = 0; a < N; ++a) rew[a] = r;
    } else { /* ORC_SCN_MT */
        /* a16 (MaterialTransport.py:113-189) */
        int32_t *load = st->load + (size_t)e * N, *zone = st->zone_load + (size_t)e * 2, *msg = st->messages + (size_t)e * 4;
        for (int i = 0; i < 4 && i < N; ++i) msg[i] = act[i] % 4;
        for (int a = 0; a < N; ++a) {
            REAL *o = obs + (size_t)a * D;
            o[0] = x[a];
            o[1] = y[a];
            o[2] = (REAL)load[a];
            o[3] = (REAL)zone[0];
            o[4] = (REAL)zone[1];
            for (int i = 0; i < 4; ++i) o[5 + i] = (REAL)msg[i];
            if (p->capability_aware) {
                o[9] = (REAL)p->torque[a];
                o[10] = R(p->agent_step[a]);
            }
        }
        REAL r;
        if (viol) {
            r = R(p->violation_reward);
            done = 1;
        } else {
            r = R(p->time_penalty);
            REAL egw = R(p->end_goal_width);
            for (int a = 0; a < N; ++a) {
                if (load[a] > 0) {
                    if (x[a] < R(-1.5) + egw) {
                        r = r + (REAL)load[a] * R(p->unload_multiplier);
                        load[a] = 0;
                    }
                } else {
                    if (x[a] > R(1.5) - egw) {
                        if (zone[1] > p->torque[a]) {
                            load[a] = p->torque[a];
                            zone[1] -= p->torque[a];
                        } else {
                            load[a] = zone[1];
                            zone[1] = 0;
                        }
                        r = r + (REAL)load[a] * R(p->load_multiplier);
                    } else if (DIST(x[a] * x[a] + y[a] * y[a]) <= RADIUS(R(p->zone1_radius))) {
                        if (zone[0] > p->torque[a]) {
                            load[a] = p->torque[a];
                            zone[0] -= p->torque[a];
                        } else {
                            load[a] = zone[0];
                            zone[0] = 0;
                        }
                        r = r + (REAL)load[a] * R(p->load_multiplier);
                    }
                }
            }
            done = steps > p->max_episode_steps;
            if (!done) {
                done = (zone[0] == 0 && zone[1] == 0);
                if (done)
                    for (int a = 0; a < N; ++a)
                        if (load[a] != 0) {
                            done = 0;
                            break;
                        }
            }
        }
        if (done) {
            remaining = zone[0] + zone[1];
            for (int a = 0; a < N; ++a) remaining += load[a];
        }
        for (int a = 0; a < N; ++a) rew[a] = r;
    }
    out->done[e] = (uint8_t)done;
    out->remaining[e] = remaining;
}

int FN(orc_step)(const orc_params *p, int E, const FN(orc_state) * st, const int32_t *actions, const FN(orc_out) * out) {
    if (p->n_agents < 1 || p->n_agents > ORC_MAXN) return -1;
    if (p->scenario == ORC_SCN_PCP && (p->num_prey < 0 || p->num_prey > ORC_MAXP)) return -2;
    if (p->scenario == ORC_SCN_ARCTIC && p->n_agents != 4) return -3;
    for (int e = 0; e < E; ++e) FN(step_env)(p, e, st, actions, out);
    return 0;
}

/* standalone entry points for unit tests of the spec'd pieces */
void FN(orc_sincos)(int n, const REAL *t, REAL *s, REAL *c) {
    for (int i = 0; i < n; ++i) SINCOS(t[i], &s[i], &c[i]);
}
void FN(orc_atan2)(int n, const REAL *y, const REAL *x, REAL *o) {
    for (int i = 0; i < n; ++i) o[i] = ATAN2(y[i], x[i]);
}
#if ORC_IS_F32
/* tests: ipm_spec_v0 on binary32 records (xi_x, xi_y, uhat_x, uhat_y) x N, exactly what the kernels' LDS records hold; the iterate
 * replaces uhat.  Returns the iteration count. */
void orc_ipm_rcp(int n, const double *v, double *r) {
    for (int i = 0; i < n; ++i) r[i] = ipm_rcp(v[i]);
}
int orc_ipm_spec_f32io(const orc_params *p, int N, float *io) {
    double xi[ORC_MAXN], yi[ORC_MAXN], ux[ORC_MAXN], uy[ORC_MAXN];
    for (int a = 0; a < N; ++a) {
        xi[a] = io[4 * a];
        yi[a] = io[4 * a + 1];
        ux[a] = io[4 * a + 2];
        uy[a] = io[4 * a + 3];
    }
    const ipm_consts kc = ipm_make_consts(p, 1);
    int it = barrier_qp_ipm_spec(&kc, N, xi, yi, ux, uy);
    for (int a = 0; a < N; ++a) {
        io[4 * a + 2] = (float)ux[a];
        io[4 * a + 3] = (float)uy[a];
    }
    return it;
}
#endif
int FN(orc_controller)(const orc_params *p, const REAL *poses /*3xN*/, const REAL *goals /*2xN*/, REAL *dxu /*2xN*/) {
    int N = p->n_agents;
    REAL cs[ORC_MAXN], ss[ORC_MAXN];
    for (int a = 0; a < N; ++a) SINCOS(poses[2 * N + a], &ss[a], &cs[a]);
    return FN(controller)(p, N, poses, poses + N, cs, ss, goals, goals + N, dxu, dxu + N);
}

#undef SINCOS
#undef ATAN2
#undef WRAP
#ifdef SINCOS_STEP
#undef SINCOS_STEP
#endif
#undef SQRT
#undef FMA
#undef DIST
#undef RADIUS
#undef R
