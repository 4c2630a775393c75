"""Independent vectorised numpy restatement of the assembly math — TEST INFRASTRUCTURE.

Written from SURVEY.md Appendix A (not from nos_oracle.c) so that it can cross-check the C
oracle: same plane order, same {upper(H) | g | cost} output.  Also provides the plain cost
function for finite-difference checks of the analytic Jacobians (g must equal half the
gradient of the robust cost under the right-perturbation R <- R Exp(dw), t <- t + dt).

Reference lines: MDM/..._analytic.cc:159-185 (6-DoF), MDM/..._analytic_3dof.cc:110-139 (3-DoF),
REM/..._analytic.cc:107-162 (reprojection), NO/loss_function.h:28-33,57-66 (losses).
"""
import numpy as np

TRI6 = [(r, c) for r in range(6) for c in range(r, 6)]
TRI3 = [(r, c) for r in range(3) for c in range(r, 3)]


def loss_eval(loss, s):
    if loss is None or loss[0] == "none":
        return s.copy(), np.ones_like(s)
    if loss[0] == "exponential":
        c1, c2 = float(loss[1]), float(loss[2])
        ex = np.exp(-c2 * s)
        return c1 - c1 * ex, 2.0 * c1 * c2 * ex
    if loss[0] == "huber":
        th = float(loss[1])
        out = s > th * th
        rr = np.sqrt(np.where(out, s, 1.0))
        return np.where(out, 2.0 * th * rr - th * th, s), np.where(out, th / rr, 1.0)
    raise ValueError(loss)


def hat(p):
    """[n,3] → [n,3,3] skew matrices."""
    n = p.shape[0]
    K = np.zeros((n, 3, 3))
    K[:, 0, 1] = -p[:, 2]
    K[:, 0, 2] = p[:, 1]
    K[:, 1, 0] = p[:, 2]
    K[:, 1, 2] = -p[:, 0]
    K[:, 2, 0] = -p[:, 1]
    K[:, 2, 1] = p[:, 0]
    return K


def _pack(H, g, cost, tri):
    return np.concatenate([[H[r, c] for r, c in tri], g, [cost]])


def ndt6_terms(planes, R, t):
    p = planes[0:3].T
    mu = planes[3:6].T
    S = planes[6:15].T.reshape(-1, 3, 3)
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    e = p @ R.T + np.asarray(t) - mu
    r = np.einsum("nij,nj->ni", S, e)
    M = -np.einsum("ij,njk->nik", R, hat(p))
    J = np.concatenate([S, np.einsum("nij,njk->nik", S, M)], axis=2)
    return r, J


def ndt6_accumulate(planes, R, t, loss=None):
    r, J = ndt6_terms(planes, R, t)
    s = np.einsum("ni,ni->n", r, r)
    rho, w = loss_eval(loss, s)
    H = np.einsum("n,nki,nkj->ij", w, J, J)
    g = np.einsum("n,nki,nk->i", w, J, r)
    return _pack(H, g, rho.sum(), TRI6)


def ndt6_cost(planes, R, t, loss=None):
    r, _ = ndt6_terms(planes, R, t)
    rho, _ = loss_eval(loss, np.einsum("ni,ni->n", r, r))
    return rho.sum()


def ndt3_terms(planes, R2, t2):
    p = planes[0:3].T
    mu = planes[3:6].T
    S = planes[6:15].T.reshape(-1, 3, 3)
    R2 = np.asarray(R2, dtype=np.float64).reshape(2, 2)
    u = p[:, :2]
    uw = u @ R2.T + np.asarray(t2)
    e = np.concatenate([uw, p[:, 2:3]], axis=1) - mu
    r = np.einsum("nij,nj->ni", S, e)
    d = np.stack([-u[:, 1], u[:, 0]], axis=1) @ R2.T
    J = np.concatenate([S[:, :, :2], np.einsum("nij,nj->ni", S[:, :, :2], d)[:, :, None]], axis=2)
    return r, J


def ndt3_accumulate(planes, R2, t2, loss=None):
    r, J = ndt3_terms(planes, R2, t2)
    s = np.einsum("ni,ni->n", r, r)
    rho, w = loss_eval(loss, s)
    H = np.einsum("n,nki,nkj->ij", w, J, J)
    g = np.einsum("n,nki,nk->i", w, J, r)
    return _pack(H, g, rho.sum(), TRI3)


def ndt3_cost(planes, R2, t2, loss=None):
    r, _ = ndt3_terms(planes, R2, t2)
    rho, _ = loss_eval(loss, np.einsum("ni,ni->n", r, r))
    return rho.sum()


def reproj_terms(planes, R, t, intr, min_depth=0.03):
    X = planes[0:3].T
    px = planes[3:5].T
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    inv_fx, inv_fy, cx, cy = intr
    Xw = X @ R.T + np.asarray(t)
    ok = ~(Xw[:, 2] < min_depth)
    z = np.where(ok, Xw[:, 2], 1.0)
    iz = 1.0 / z
    r = np.stack([Xw[:, 0] * iz - inv_fx * (px[:, 0] - cx), Xw[:, 1] * iz - inv_fy * (px[:, 1] - cy)], axis=1)
    n = X.shape[0]
    dK = np.zeros((n, 2, 3))
    dK[:, 0, 0] = iz
    dK[:, 0, 2] = -Xw[:, 0] * iz * iz
    dK[:, 1, 1] = iz
    dK[:, 1, 2] = -Xw[:, 1] * iz * iz
    M = -np.einsum("ij,njk->nik", R, hat(X))
    J = np.concatenate([dK, np.einsum("nij,njk->nik", dK, M)], axis=2)
    r = r * ok[:, None]
    J = J * ok[:, None, None]
    return r, J


def reproj_accumulate(planes, R, t, intr, loss=None, min_depth=0.03):
    r, J = reproj_terms(planes, R, t, intr, min_depth)
    s = np.einsum("ni,ni->n", r, r)
    rho, w = loss_eval(loss, s)
    H = np.einsum("n,nki,nkj->ij", w, J, J)
    g = np.einsum("n,nki,nk->i", w, J, r)
    return _pack(H, g, rho.sum(), TRI6)


def reproj_accumulate_simd_class(planes, R, t, intr, loss=None):
    """The fp32 class's rules in fp64 (REM/reprojection_error_minimizer_analytic_simd.cc:55-138): residual and Jacobian
    of EVERY correspondence from its real depth, mask = depth > 0 multiplying the weight only (:66,92), the loss of a
    masked correspondence still added to the cost (:134).  (The tail drop to floor(N/8)*8 is the caller's.)"""
    X = planes[0:3].T
    px = planes[3:5].T
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    inv_fx, inv_fy, cx, cy = intr
    Xw = X @ R.T + np.asarray(t)
    inside = Xw[:, 2] > 0.0
    iz = 1.0 / Xw[:, 2]
    r = np.stack([Xw[:, 0] * iz - inv_fx * (px[:, 0] - cx), Xw[:, 1] * iz - inv_fy * (px[:, 1] - cy)], axis=1)
    n = X.shape[0]
    dK = np.zeros((n, 2, 3))
    dK[:, 0, 0] = iz
    dK[:, 0, 2] = -Xw[:, 0] * iz * iz
    dK[:, 1, 1] = iz
    dK[:, 1, 2] = -Xw[:, 1] * iz * iz
    M = -np.einsum("ij,njk->nik", R, hat(X))
    J = np.concatenate([dK, np.einsum("nij,njk->nik", dK, M)], axis=2)
    s = np.einsum("ni,ni->n", r, r)
    rho, w = loss_eval(loss, s)
    w = w * inside
    H = np.einsum("n,nki,nkj->ij", w, J, J)
    g = np.einsum("n,nki,nk->i", w, J, r)
    return _pack(H, g, rho.sum(), TRI6)


def reproj_cost(planes, R, t, intr, loss=None, min_depth=0.03):
    r, _ = reproj_terms(planes, R, t, intr, min_depth)
    rho, _ = loss_eval(loss, np.einsum("ni,ni->n", r, r))
    return rho.sum()


def exp_so3(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    K = hat(w[None, :])[0]
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / (th * th) * K @ K
