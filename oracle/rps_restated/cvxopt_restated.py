"""ORACLE -- test infrastructure only.  `qp(P, q, G, h)` as cvxopt's `solvers.qp` computes it for a dense linear-inequality QP
(the only form rps' barrier certificates use), restated FROM MEMORY of cvxopt 1.3's `coneqp` -- cvxopt is absent from this
image and from /root/reference, so this is PARITY UNPINNED against the real package.

Restated: the default starting point (the least-squares KKT system with the identity scaling, s and z shifted into the cone by
1 + max_step when they are not strictly inside), Mehrotra's predictor-corrector with Nesterov-Todd scaling -- for the linear
cone the plain (s, z) iteration --, step 0.99 to the boundary, sigma = clip(1 - step + dsdz / gap * step^2, 0, 1)^3, no
iterative refinement (cvxopt's default for a problem without second-order / semidefinite blocks), stop when
pres, dres <= feastol and (gap <= abstol or relgap <= reltol), or at maxiters.  Float64 numpy, dense Cholesky for the KKT
systems ('chol2' is cvxopt's default here; the factorisation differs in rounding only).

`options` mirrors `cvxopt.solvers.options`; rps sets reltol = feastol = 1e-2, maxiters = 50 at import
(rps/utilities/barrier_certificates.py).  oracle/oracle_core.h barrier_qp_ipm is the same iteration in C.
"""
import numpy as np

options = {"show_progress": False, "abstol": 1e-7, "reltol": 1e-6, "feastol": 1e-7, "maxiters": 100}
STEP = 0.99


def _chol_solve(K, b):
    L = np.linalg.cholesky(K)
    y = np.linalg.solve(L, b)            # (n <= 32: the triangular structure is not worth a scipy dependency)
    return np.linalg.solve(L.T, y)


def qp(P, q, G, h, opts=None):
    """min 1/2 x'Px + q'x  s.t.  Gx <= h.  Returns {'x', 's', 'z', 'iterations', 'status', 'gap'} (numpy arrays)."""
    o = dict(options)
    if opts:
        o.update(opts)
    P = np.asarray(P, np.float64)
    q = np.asarray(q, np.float64).reshape(-1)
    G = np.asarray(G, np.float64)
    h = np.asarray(h, np.float64).reshape(-1)
    n, m = q.size, h.size
    if m == 0:
        x = np.linalg.solve(P, -q)
        return {"x": x, "s": h.copy(), "z": h.copy(), "iterations": 0, "status": "optimal", "gap": 0.0}
    resx0 = max(1.0, np.sqrt(q @ q))
    resz0 = max(1.0, np.sqrt(h @ h))
    # default starting point: [P G'; G -I][x; z] = [-q; h]
    x = _chol_solve(P + G.T @ G, -q + G.T @ h)
    z = G @ x - h
    s = -z
    nrms = np.sqrt(s @ s)
    ts = np.max(-s)
    if ts >= -1e-8 * max(nrms, 1.0):
        s = s + (1.0 + ts)
    nrmz = np.sqrt(z @ z)
    tz = np.max(-z)
    if tz >= -1e-8 * max(nrmz, 1.0):
        z = z + (1.0 + tz)
    gap = s @ z
    status = "unknown"
    iters = 0
    while True:
        rx = P @ x + q
        f0 = 0.5 * (x @ rx + x @ q)
        rx = rx + G.T @ z
        resx = np.sqrt(rx @ rx)
        rz = s + G @ x - h
        resz = np.sqrt(rz @ rz)
        pcost = f0
        dcost = f0 + z @ rz - gap
        relgap = gap / -pcost if pcost < 0.0 else gap / dcost if dcost > 0.0 else None
        pres, dres = resz / resz0, resx / resx0
        if pres <= o["feastol"] and dres <= o["feastol"] and (gap <= o["abstol"] or (relgap is not None and relgap <= o["reltol"])):
            status = "optimal"
            break
        if iters == o["maxiters"]:
            break
        K = P + G.T @ ((z / s)[:, None] * G)
        mu = gap / m
        sigma = 0.0
        dsa_dza = np.zeros(m)
        for i in (0, 1):
            rc = -s * z + sigma * mu - (dsa_dza if i else 0.0)
            dx = _chol_solve(K, -rx - G.T @ ((rc + z * rz) / s))
            ds = -rz - G @ dx
            dz = (rc - z * ds) / s
            dsdz = ds @ dz
            t = max(0.0, np.max(-ds / s), np.max(-dz / z))
            if i == 0:
                step = 1.0 if t == 0.0 else min(1.0, 1.0 / t)
                sigma = min(1.0, max(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** 3
                dsa_dza = ds * dz
            else:
                step = 1.0 if t == 0.0 else min(1.0, STEP / t)
        x = x + step * dx
        s = s + step * ds
        z = z + step * dz
        gap = s @ z
        iters += 1
    return {"x": x, "s": s, "z": z, "iterations": iters, "status": status, "gap": gap}
