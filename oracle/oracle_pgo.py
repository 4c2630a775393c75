"""CPU restatement of pose-graph optimisation — TEST INFRASTRUCTURE.

Residual: nonlinear_optimizer/pose_graph_optimizer/ceres_cost_functor.h:17-53 (plain) and :55-98
(switchable: residual scaled by s, seventh residual (1 - s) * 1e-9).  The reference evaluates it through
Ceres autodiff only (its analytic Solve is an empty loop, pose_graph_optimizer_analytic.cc:21-42), and no
captured PGO run exists under results/, so parity for this row is UNPINNED against reference outputs; this
module pins the GPU path instead by (i) restating the residual literally, (ii) checking the analytic Jacobians
against central differences, (iii) assembling the sparse normal matrix explicitly and solving it directly
(scipy.sparse) where the GPU path uses a matrix-free preconditioned CG.

Unknown ordering of vectors: 6 planes of n_poses (dp_x, dp_y, dp_z, dw_x, dw_y, dw_z), then one switch entry
per constraint — the layout of nos_pgo_get_vector.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

C_SWITCH = 1e-9  # ceres_cost_functor.h:93


def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qrot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def qexp(w):
    th = np.linalg.norm(w)
    if th < 1e-6:
        return np.array([1.0, 0.5 * w[0], 0.5 * w[1], 0.5 * w[2]])
    k = np.sin(0.5 * th) / th
    return np.array([np.cos(0.5 * th), k * w[0], k * w[1], k * w[2]])


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def quat_from_matrix(R):
    """Eigen's Quaternion(Matrix3) (used when constraints are built from Pose objects)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([w, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[1 + i] = 0.5 * s
    s = 0.5 / s
    q[0] = (R[k, j] - R[j, k]) * s
    q[1 + j] = (R[j, i] + R[i, j]) * s
    q[1 + k] = (R[k, i] + R[i, k]) * s
    return q


def edge_residual(pr, qr, pq, qq, tm, qm):
    """6-vector of ceres_cost_functor.h:44-51."""
    rt = (pq - pr) - qrot(qr) @ tm
    e = qmul(qmul(qconj(qq), qr), qm)
    return np.concatenate([rt, 2.0 * e[1:]]), e


def edge_jacobians(qr, tm, qm, e):
    """(J_r, J_q): 6x6 each, w.r.t. (dp, dw) with p <- p + dp, q <- q (x) Exp(dw)."""
    Ep = e[0] * np.eye(3) + hat(e[1:])
    Em = -e[0] * np.eye(3) + hat(e[1:])
    Jr = np.zeros((6, 6))
    Jq = np.zeros((6, 6))
    Jr[:3, :3] = -np.eye(3)
    Jr[:3, 3:] = qrot(qr) @ hat(tm)
    Jr[3:, 3:] = Ep @ qrot(qm).T
    Jq[:3, :3] = np.eye(3)
    Jq[3:, 3:] = Em
    return Jr, Jq


class Graph:
    def __init__(self, poses, ref, qry, meas, switch_init=None, switch_free=None, fixed=None):
        self.poses = np.array(poses, dtype=np.float64).reshape(-1, 7)
        self.ref = np.asarray(ref, dtype=np.int64)
        self.qry = np.asarray(qry, dtype=np.int64)
        self.meas = np.array(meas, dtype=np.float64).reshape(-1, 7)
        m = self.ref.size
        self.sw = np.ones(m) if switch_init is None else np.array(switch_init, dtype=np.float64)
        self.sw_free = np.zeros(m, dtype=bool) if switch_free is None else np.asarray(switch_free, dtype=bool)
        self.fixed = np.zeros(self.poses.shape[0], dtype=bool) if fixed is None else np.asarray(fixed, dtype=bool)

    @property
    def n(self):
        return self.poses.shape[0]

    @property
    def m(self):
        return self.ref.size

    def cost(self):
        c = 0.0
        for e in range(self.m):
            r, _ = edge_residual(self.poses[self.ref[e], :3], self.poses[self.ref[e], 3:], self.poses[self.qry[e], :3],
                                 self.poses[self.qry[e], 3:], self.meas[e, :3], self.meas[e, 3:])
            s = self.sw[e]
            c += s * s * float(r @ r)
            if self.sw_free[e]:
                c += (1 - s) ** 2 * C_SWITCH ** 2
        return c

    def _index(self, pose, k):
        return k * self.n + pose

    def linearize(self):
        """→ (H sparse csr [6n+m, 6n+m], g [6n+m], cost).  Fixed poses get identity rows; switches that are
        not free get identity rows."""
        n, m = self.n, self.m
        dim = 6 * n + m
        rows, cols, vals = [], [], []
        g = np.zeros(dim)
        cost = 0.0
        for e in range(m):
            ir, iq = int(self.ref[e]), int(self.qry[e])
            r, eq = edge_residual(self.poses[ir, :3], self.poses[ir, 3:], self.poses[iq, :3], self.poses[iq, 3:],
                                  self.meas[e, :3], self.meas[e, 3:])
            Jr, Jq = edge_jacobians(self.poses[ir, 3:], self.meas[e, :3], self.meas[e, 3:], eq)
            s = self.sw[e]
            free = bool(self.sw_free[e])
            # full 7 x 13 Jacobian [s Jr | s Jq | r ; 0 0 -c]
            J = np.zeros((7, 13))
            J[:6, :6] = 0.0 if self.fixed[ir] else s * Jr
            J[:6, 6:12] = 0.0 if self.fixed[iq] else s * Jq
            if free:
                J[:6, 12] = r
                J[6, 12] = -C_SWITCH
            f = np.concatenate([s * r, [C_SWITCH * (1 - s) if free else 0.0]])
            cost += float(f @ f)
            idx = [self._index(ir, k) for k in range(6)] + [self._index(iq, k) for k in range(6)] + [6 * n + e]
            Hl = J.T @ J
            gl = J.T @ f
            for a in range(13):
                g[idx[a]] += gl[a]
                for b in range(13):
                    if Hl[a, b] != 0.0:
                        rows.append(idx[a])
                        cols.append(idx[b])
                        vals.append(Hl[a, b])
        H = sp.coo_matrix((vals, (rows, cols)), shape=(dim, dim)).tocsr()
        # identity rows for fixed poses / fixed switches
        diag_fix = np.zeros(dim)
        for i in np.nonzero(self.fixed)[0]:
            for k in range(6):
                diag_fix[self._index(int(i), k)] = 1.0
        for e in np.nonzero(~self.sw_free)[0]:
            diag_fix[6 * n + int(e)] = 1.0
        H = H + sp.diags(diag_fix)
        return H.tocsr(), g, cost

    def damped(self, H, lam):
        return (H + lam * sp.diags(H.diagonal())).tocsc()

    def solve_step(self, H, g, lam):
        return spla.spsolve(self.damped(H, lam), -g)

    def retract(self, dx):
        n = self.n
        for i in range(n):
            if self.fixed[i]:
                continue
            d = np.array([dx[self._index(i, k)] for k in range(6)])
            self.poses[i, :3] += d[:3]
            q = qmul(self.poses[i, 3:], qexp(d[3:]))
            self.poses[i, 3:] = q / np.linalg.norm(q)
        for e in range(self.m):
            if self.sw_free[e]:
                self.sw[e] += dx[6 * n + e]

    def optimize(self, max_iterations=40, gradient_tolerance=1e-6, parameter_tolerance=1e-6, lam0=1e-3):
        """The LM loop of the reference's analytic solvers (MDM/..._analytic_simd.cc:30-108) on this problem:
        always apply the step, lambda x2 / x0.6 on the cost, clamp [1e-6, 1e-2], convergence after the update."""
        lam, prev = lam0, np.finfo(np.float64).max
        it = 0
        hist = []
        for it in range(max_iterations):
            H, g, cost = self.linearize()
            dx = self.solve_step(H, g, lam)
            self.retract(dx)
            hist.append((cost, float(np.linalg.norm(g)), float(np.linalg.norm(dx))))
            if np.linalg.norm(dx) < parameter_tolerance or np.linalg.norm(g) < gradient_tolerance:
                break
            lam = min(max(lam * (2.0 if cost > prev else 0.6), 1e-6), 1e-2)
            prev = cost
        return it, hist


def reference_test_scene():
    """The reference's PGO demo (nonlinear_optimizer/pose_graph_optimizer/tests/simple_optimization_test.cc:19-122):
    80-pose square loop with 0.2 m steps, deterministic +-0.08 m position noise on every pose but the first, 79
    odometry constraints + 4 loop constraints from the TRUE poses, the last loop constraint replaced by identity
    (an outlier).  → (true poses [80,7], noisy poses [80,7], ref, qry, meas [83,7], switch_free [83])."""
    pos = np.zeros((80, 3))
    x = y = z = 0.0
    for i in range(20):
        pos[i] = (x, y, z)
        x += 0.2
        z += 0.2
    for i in range(20, 40):
        y += 0.2
        z += 0.2
        pos[i] = (x, y, z)
    for i in range(40, 60):
        x -= 0.2
        z -= 0.2
        pos[i] = (x, y, z)
    for i in range(60, 80):
        y -= 0.2
        z -= 0.2
        pos[i] = (x, y, z)
    true = np.zeros((80, 7))
    true[:, :3] = pos
    true[:, 3] = 1.0
    noisy = true.copy()
    noisy[0, :3] = 0.0  # noisy_poses.push_back(Pose::Identity())
    for i in range(1, 80):
        noisy[i, i % 3] += (1 if i % 2 else -1) * 0.08
    pairs = [(i, i + 1) for i in range(79)] + [(18, 21), (38, 42), (57, 61), (77, 2)]
    ref = np.array([a for a, _ in pairs], dtype=np.int32)
    qry = np.array([b for _, b in pairs], dtype=np.int32)
    meas = np.zeros((83, 7))
    meas[:, 3] = 1.0
    for e, (a, b) in enumerate(pairs):
        meas[e, :3] = true[b, :3] - true[a, :3]  # identity rotations: R_a^T (p_b - p_a)
    meas[82, :3] = 0.0  # outlier: setIdentity()
    free = np.zeros(83, dtype=np.uint8)
    free[79:] = 1  # loop constraints carry a free switch (pose_graph_optimizer_ceres.cc:31-36)
    return true, noisy, ref, qry, meas, free


def random_graph(n, extra_edges_per_pose=3, seed=0, noise_t=0.05, noise_r=0.02, meas_noise_t=0.01, meas_noise_r=0.005):
    """Synthetic graph of the configs[4] shape: a smooth 3-D trajectory with odometry constraints i → i+1 and
    `extra_edges_per_pose` loop constraints to random poses a short way ahead.  → dict."""
    rng = np.random.default_rng(seed)
    true = np.zeros((n, 7))
    q = np.array([1.0, 0, 0, 0])
    p = np.zeros(3)
    for i in range(n):
        true[i, :3] = p
        true[i, 3:] = q
        dq = qexp(rng.normal(scale=0.05, size=3))
        q = qmul(q, dq)
        q /= np.linalg.norm(q)
        p = p + qrot(q) @ np.array([0.5, 0.0, 0.0]) + rng.normal(scale=0.02, size=3)
    pairs = [(i, i + 1) for i in range(n - 1)]
    for i in range(n):
        for _ in range(extra_edges_per_pose):
            j = i + int(rng.integers(2, 40))
            if j < n:
                pairs.append((i, j))
    ref = np.array([a for a, _ in pairs], dtype=np.int32)
    qry = np.array([b for _, b in pairs], dtype=np.int32)
    meas = np.zeros((len(pairs), 7))
    for e, (a, b) in enumerate(pairs):
        Ra = qrot(true[a, 3:])
        t = Ra.T @ (true[b, :3] - true[a, :3]) + rng.normal(scale=meas_noise_t, size=3)
        qm = qmul(qmul(qconj(true[a, 3:]), true[b, 3:]), qexp(rng.normal(scale=meas_noise_r, size=3)))
        meas[e, :3] = t
        meas[e, 3:] = qm / np.linalg.norm(qm)
    init = true.copy()
    for i in range(1, n):
        init[i, :3] += rng.normal(scale=noise_t, size=3)
        qi = qmul(init[i, 3:], qexp(rng.normal(scale=noise_r, size=3)))
        init[i, 3:] = qi / np.linalg.norm(qi)
    fixed = np.zeros(n, dtype=np.uint8)
    fixed[0] = 1
    return {"true": true, "init": init, "ref": ref, "qry": qry, "meas": meas, "fixed": fixed}
