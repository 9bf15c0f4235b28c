"""ORACLE -- restated `rps.utilities.barrier_certificates` (parity vs real rps + cvxopt unpinned).

Spec: SURVEY.md Appendix A.6.  Reference call sites: utilities/controller.py:2,13-16,23.

Upstream builds  min ||u - dxi||^2  s.t.  -2 e_ij.u_i + 2 e_ij.u_j <= gamma * h_ij^3
and hands it to cvxopt's interior-point `qp` at loose tolerances (reltol 1e-2), so its
answer is an approximate iterate no other solver reproduces.  cvxopt is absent here.  Two solvers
stand in for it, chosen by the module attribute QP_SOLVER (set by the test harness from the config key
`barrier_solver`):
  "exact"  (sim_spec_v0's default): the EXACT projection, computed by Hildreth's dual coordinate ascent
           in a fixed constraint order (the order the HIP kernel can run pair-parallel: the XOR
           1-factorisation of the lane group).  `solve_pair_qp` below is the float64 statement of
           that spec; tests/test_qp.py checks it against an independent active-set solution
           (scipy NNLS / Lawson-Hanson LDP).
  "cvxopt" : upstream's own formulation (the matrices below are built as rps builds them) handed to the
           restated interior-point `qp` of oracle/rps_restated/cvxopt_restated.py at rps' options
           (reltol = feastol = 1e-2, maxiters 50): what `barrier_solver: cvxopt` of the product computes.
"""
import numpy as np
from rps.utilities.transformations import *   # upstream star-imports these; controller.py relies on it

import cvxopt_restated  # noqa: E402  (oracle/rps_restated is on sys.path wherever this package is importable)

QP_SOLVER = "exact"
CVXOPT_OPTIONS = {"reltol": 1e-2, "feastol": 1e-2, "maxiters": 50}   # rps sets these at import (Appendix A.6)
QP_RTOL_F64 = 5e-12
QP_MAX_SWEEPS_F64 = 200


def group_width(N):
    gw = 2
    while gw < N:
        gw *= 2
    return gw


def pair_order(N):
    """Constraint order of sim_spec_v0: for k = 1..GW-1, pairs (a, a^k) with a < a^k < N."""
    gw = group_width(N)
    order = []
    for k in range(1, gw):
        for a in range(N):
            p = a ^ k
            if a < p < N:
                order.append((a, p))
    return order


def solve_pair_qp(uhat, x, beta, rtol=QP_RTOL_F64, max_sweeps=QP_MAX_SWEEPS_F64, magnitude_limit=0.2):
    """Projection of uhat (2xN) onto { u : e_ij.(u_j - u_i) <= beta_ij for all i<j },
    e_ij = x_i - x_j.  (Upstream's row  -2e.u_i + 2e.u_j <= b  divided by two; beta = b/2.)
    beta is a dict {(i,j): value}.  Returns (u, sweeps).

    Hildreth sweeps in the XOR-factorisation pair order, with a vector (delta-squared) extrapolation
    of the multipliers after sweeps 3, 7, 11, ... (see oracle/oracle_core.h barrier_qp, the same
    algorithm in C)."""
    N = uhat.shape[1]
    gw = group_width(N)
    u = uhat.copy()
    order = pair_order(N)
    e = {(i, j): (x[0, i] - x[0, j], x[1, i] - x[1, j]) for (i, j) in order}
    n2 = {pr: 2.0 * (e[pr][0] * e[pr][0] + e[pr][1] * e[pr][1]) for pr in order}
    order = [pr for pr in order if n2[pr] > 0.0]
    rn2 = {pr: 1.0 / n2[pr] for pr in order}
    mu = {pr: 0.0 for pr in order}
    muA = dict(mu)
    muB = dict(mu)
    sweeps = 0
    while True:
        maxchg = 0.0
        muA, muB = muB, dict(mu)
        for (i, j) in order:
            ex, ey = e[(i, j)]
            r = rn2[(i, j)]
            mn = (mu[(i, j)] - (0.5 * (2.0 * beta[(i, j)])) * r) + ((ey * r) * (u[1, j] - u[1, i])) \
                + ((ex * r) * (u[0, j] - u[0, i]))
            mn = max(0.0, mn)
            delta = mn - mu[(i, j)]
            mu[(i, j)] = mn
            u[0, i] += delta * ex
            u[1, i] += delta * ey
            u[0, j] -= delta * ex
            u[1, j] -= delta * ey
            maxchg = max(maxchg, abs(delta) * max(abs(ex), abs(ey)))
        sweeps += 1
        umax = max(magnitude_limit, float(np.abs(u).max()))
        if maxchg <= rtol * umax or sweeps >= max_sweeps:
            break
        if (sweeps & 3) == 3:
            pa, pb = [0.0] * gw, [0.0] * gw           # per robot over its partners in round order, then the lane butterfly
            for a in range(N):
                for k in range(1, gw):
                    q = a ^ k
                    pr = (min(a, q), max(a, q))
                    if q >= N or pr not in mu:
                        continue
                    d1, d2 = muB[pr] - muA[pr], mu[pr] - muB[pr]
                    dd = d2 - d1
                    pa[a] += dd * d2
                    pb[a] += dd * dd
            stride = 1
            while stride < gw:
                for a in range(0, gw, 2 * stride):
                    pa[a] += pa[a + stride]
                    pb[a] += pb[a + stride]
                stride *= 2
            ga, gb = pa[0], pb[0]
            gam = ga / gb if (gb > 0.0 and ga < 0.0 and -ga < 32.0 * gb) else 0.0
            for pr in order:
                mu[pr] = max(0.0, mu[pr] - gam * (mu[pr] - muB[pr]))
            u = uhat.copy()
            for a in range(N):
                for k in range(1, gw):
                    q = a ^ k
                    if q >= N:
                        continue
                    pr = (min(a, q), max(a, q))
                    if pr not in mu:
                        continue
                    s = 1.0 if a < q else -1.0
                    u[0, a] += mu[pr] * (s * e[pr][0])
                    u[1, a] += mu[pr] * (s * e[pr][1])
    return u, sweeps


def _make_certificate(barrier_gain, unsafe_barrier_gain, safety_radius, magnitude_limit):
    def f(dxi, x):
        N = dxi.shape[1]
        if QP_SOLVER == "cvxopt":
            return _certificate_as_upstream(dxi, x, barrier_gain, unsafe_barrier_gain, safety_radius, magnitude_limit)
        beta = {}
        for i in range(N - 1):
            for j in range(i + 1, N):
                error = x[:, i] - x[:, j]
                h = (error[0] * error[0] + error[1] * error[1]) - np.power(safety_radius, 2)
                if h >= 0 or unsafe_barrier_gain is None:
                    b = barrier_gain * np.power(h, 3)
                else:
                    b = unsafe_barrier_gain * np.power(h, 3)
                beta[(i, j)] = 0.5 * b
        # Threshold control inputs before QP (in place, as upstream)
        norms = np.linalg.norm(dxi, 2, 0)
        idxs_to_normalize = (norms > magnitude_limit)
        dxi[:, idxs_to_normalize] *= magnitude_limit / norms[idxs_to_normalize]
        u, _ = solve_pair_qp(dxi, x, beta, magnitude_limit=magnitude_limit)
        return u
    return f


def _certificate_as_upstream(dxi, x, barrier_gain, unsafe_barrier_gain, safety_radius, magnitude_limit):
    """The body of rps' certificate closures as recalled (Appendix A.6): dense A (rows -2e at robot i, +2e at robot j), b, H = 2I,
    f = -2 vec_F(dxi) after the in-place thresholding, then `qp(H, f, A, b)['x']` reshaped column-major."""
    N = dxi.shape[1]
    num_constraints = N * (N - 1) // 2
    A = np.zeros((num_constraints, 2 * N))
    b = np.zeros(num_constraints)
    H = 2 * np.identity(2 * N)
    count = 0
    for i in range(N - 1):
        for j in range(i + 1, N):
            error = x[:, i] - x[:, j]
            h = (error[0] * error[0] + error[1] * error[1]) - np.power(safety_radius, 2)
            A[count, (2 * i, (2 * i + 1))] = -2 * error
            A[count, (2 * j, (2 * j + 1))] = 2 * error
            if h >= 0 or unsafe_barrier_gain is None:
                b[count] = barrier_gain * np.power(h, 3)
            else:
                b[count] = unsafe_barrier_gain * np.power(h, 3)
            count += 1
    norms = np.linalg.norm(dxi, 2, 0)
    idxs_to_normalize = (norms > magnitude_limit)
    dxi[:, idxs_to_normalize] *= magnitude_limit / norms[idxs_to_normalize]
    f = -2 * np.reshape(dxi, 2 * N, order='F')
    result = cvxopt_restated.qp(H, f, A, b, CVXOPT_OPTIONS)['x']
    return np.reshape(result, (2, -1), order='F')


def create_single_integrator_barrier_certificate(barrier_gain=100, safety_radius=0.17, magnitude_limit=0.2):
    return _make_certificate(barrier_gain, None, safety_radius, magnitude_limit)


def create_single_integrator_barrier_certificate2(barrier_gain=100, unsafe_barrier_gain=1e6,
                                                  safety_radius=0.17, magnitude_limit=0.2):
    return _make_certificate(barrier_gain, unsafe_barrier_gain, safety_radius, magnitude_limit)
