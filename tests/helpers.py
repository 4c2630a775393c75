"""Shared helpers for the parity tests."""
import numpy as np

TRI6 = [(r, c) for r in range(6) for c in range(r, 6)]
TRI3 = [(r, c) for r in range(3) for c in range(r, 3)]


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def unpack(out, dim):
    """out = {upper(H) | g | cost} → (H full symmetric, g, cost)."""
    tri = TRI6 if dim == 6 else TRI3
    H = np.zeros((dim, dim))
    for k, (r, c) in enumerate(tri):
        H[r, c] = H[c, r] = out[k]
    g = np.array(out[len(tri):len(tri) + dim])
    return H, g, float(out[len(tri) + dim])


def assert_normal_equations_close(got, want, dim, rtol):
    """Scale-aware comparison of {H upper | g | cost}.

    Off-diagonal sums cancel, so entries are compared against the Cauchy-Schwarz scale
    sqrt(H_ii H_jj) (and sqrt(H_ii * sum w r^2) for g) instead of their own magnitude.
    """
    Hg, gg, cg = unpack(got, dim)
    Hw, gw, cw = unpack(want, dim)
    d = np.sqrt(np.maximum(np.diag(Hw), 0.0))
    scale_H = np.outer(d, d) + 1e-300
    err_H = np.max(np.abs(Hg - Hw) / scale_H)
    assert err_H <= rtol, "H mismatch: scaled err %.3e > %.1e" % (err_H, rtol)
    # |g_i| <= sqrt(H_ii) * sqrt(sum w r^2); use max(|g|) as a floor for the second factor
    gscale = d * max(np.max(np.abs(gw) / (d + 1e-300)), 1e-300) + 1e-300
    err_g = np.max(np.abs(gg - gw) / gscale)
    assert err_g <= rtol, "g mismatch: scaled err %.3e > %.1e" % (err_g, rtol)
    err_c = abs(cg - cw) / max(abs(cw), 1e-300)
    assert err_c <= rtol, "cost mismatch: rel err %.3e > %.1e" % (err_c, rtol)


def pose_delta(Ra, ta, Rb, tb):
    """(max |dt|, max |dq|) between two poses; quaternion sign-aligned."""
    from oracle import loader
    qa = loader.quat_from_matrix(np.asarray(Ra).reshape(-1))
    qb = loader.quat_from_matrix(np.asarray(Rb).reshape(-1))
    if np.dot(qa, qb) < 0:
        qb = -qb
    return float(np.max(np.abs(np.asarray(ta) - np.asarray(tb)))), float(np.max(np.abs(qa - qb)))


def reference_reprojection_scene():
    """The reference's reprojection test scene, REM/tests/simple_optimization_test.cc:43-61,115-160:
    30x21 planar grid at z = 3 (loop variables accumulated in floating point exactly as there),
    exact projections through true_pose^-1.  Returns (planes [5, 630], (fx, fy, cx, cy), R_true, t_true)."""
    pts = []
    x = -1.5
    while x <= 1.5:
        y = -1.0
        while y <= 1.0:
            pts.append((x, y, 3.0))
            y += 0.1
        x += 0.1
    pts = np.array(pts)
    fx = fy = 525.0
    cx, cy = 320.0, 240.0
    c, s = np.cos(0.1), np.sin(0.1)
    Rt = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    tt = np.array([-0.1, 0.123, -0.5])
    q = (Rt.T @ (pts - tt).T).T
    inv_z = 1.0 / q[:, 2]
    u = fx * q[:, 0] * inv_z + cx
    v = fy * q[:, 1] * inv_z + cy
    planes = np.stack([pts[:, 0], pts[:, 1], pts[:, 2], u, v])
    return planes, (fx, fy, cx, cy), Rt, tt
