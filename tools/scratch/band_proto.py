"""CPU prototype of the two-stage reduction behind trd_band_kernel (csrc/eig_tridiag.hip):

  stage 1  dense -> band of semi-bandwidth b by block Householder (panel QR of the b columns below the band, two-sided
           rank-2b update of the trailing matrix), in the data flow of the GPU kernel: ONE exchange per PANEL -- the
           trailing matrix is touched once per panel (pending update of panel p-1 applied, then Y = A V_p), the next
           panel's columns travel in their not-yet-updated state and every workgroup applies the pending update to
           them on its own;
  stage 2  band -> tridiagonal by Householder bulge chasing (one column at a time, windows of b rows), the reflectors
           logged by time step: sweep k's step m runs at time lag*k + m, steps of one time are independent;
  back     eigenvectors of T -> Q2 (logged reflectors, reverse time order) -> Q1 (panel reflectors, reverse order).

Checked against LAPACK on random, graded (Gram-like), rank-deficient and clustered matrices.
"""
import sys

import numpy as np


def house(x):
    """LAPACK dlarfg: H x = beta e_1, H = I - tau v v^T, v[0] = 1."""
    alpha = x[0]
    sigma = float(np.dot(x[1:], x[1:]))
    v = np.zeros_like(x)
    v[0] = 1.0
    if sigma == 0.0:
        return v, 0.0, alpha
    beta = -np.copysign(np.sqrt(alpha * alpha + sigma), alpha)
    tau = (beta - alpha) / beta
    v[1:] = x[1:] / (alpha - beta)
    return v, tau, beta


def stage1(A, b):
    """Returns the band matrix (dense storage), reflectors V (row g: v_g, pivot at g + b), tau."""
    n = A.shape[0]
    A = A.copy()
    Vh = np.zeros((n, n))
    taus = np.zeros(n)
    band = np.zeros((n, n))
    pending = None  # (V, W) of the previous panel, not yet applied to A
    p = 0
    while True:
        c0 = p * b
        if c0 >= n:
            break
        bw = min(b, n - c0)
        # the panel's columns as they travel (updates through panel p-2), then the pending update applied locally
        X = A[:, c0:c0 + bw].copy()
        if pending is not None:
            Vp, Wp = pending
            X -= Vp @ Wp[c0:c0 + bw].T + Wp @ Vp[c0:c0 + bw].T
        r0 = c0 + b  # first row below the band for this panel
        # the diagonal block and everything of the panel inside the band is final now
        band[c0:, c0:c0 + bw] = 0.0
        band[c0:min(r0, n), c0:c0 + bw] = X[c0:min(r0, n)]
        if r0 >= n - 1 + 1 and r0 >= n:  # nothing below the band
            if pending is not None:
                Vp, Wp = pending
                A -= Vp @ Wp.T + Wp @ Vp.T
                pending = None
            p += 1
            continue
        # QR of X[r0:, :] by bw Householder steps
        V = np.zeros((n, bw))
        tau = np.zeros(bw)
        R = X[r0:].copy()
        m = n - r0
        steps = min(bw, m - 1) if m > 1 else 0
        for t in range(bw):
            if t < m - 1 + 0 and t < m:
                if t < m - 1:
                    v, tt, beta = house(R[t:, t])
                else:
                    v, tt, beta = np.array([1.0]), 0.0, R[t, t]
                V[r0 + t:, t] = v
                tau[t] = tt
                if tt != 0.0:
                    R[t:, t:] -= tt * np.outer(v, v @ R[t:, t:])
                R[t + 1:, t] = 0.0
        band[r0:min(r0 + bw, n), c0:c0 + bw] = np.triu(R[:min(bw, m)])
        # T factor (forward, columnwise): Q = H_0 H_1 .. = I - V T V^T
        T = np.zeros((bw, bw))
        for a in range(bw):
            T[a, a] = tau[a]
            if a > 0:
                T[:a, a] = -tau[a] * (T[:a, :a] @ (V[:, :a].T @ V[:, a]))
        # tile pass: pending update, then Y = A V
        if pending is not None:
            Vp, Wp = pending
            A -= Vp @ Wp.T + Wp @ Vp.T
        Y = A @ V
        M = T.T @ (V.T @ Y) @ T
        W = Y @ T - 0.5 * V @ M
        pending = (V, W)
        for t in range(bw):
            g = c0 + t
            if g < n:
                Vh[g] = V[:, t]
                taus[g] = tau[t]
        p += 1
    band = np.tril(band) + np.tril(band, -1).T
    return band, Vh, taus


def stage2(B, b, lag=None):
    """Band (semi-bandwidth b) -> tridiagonal.  Returns d, e and the reflector log [(time, row0, v, tau)]."""
    n = B.shape[0]
    B = B.copy()
    lag = lag or 3
    log = []
    for k in range(n - 2):
        col, rs, m = k, k + 1, 0
        while rs < n - 1:
            rows = slice(rs, min(rs + b, n))
            x = B[rows, col].copy()
            if len(x) > 1 and np.any(x[1:] != 0.0):
                v, tau, beta = house(x)
                if tau != 0.0:
                    B[rows, :] -= tau * np.outer(v, v @ B[rows, :])
                    B[:, rows] -= tau * np.outer(B[:, rows] @ v, v)
                log.append((lag * k + m, rs, v, tau))
            col, rs, m = rs, rs + b, m + 1
    d = np.diag(B).copy()
    e = np.diag(B, -1).copy()
    off = B - np.diag(d) - np.diag(e, -1) - np.diag(e, 1)
    return d, e, log, float(np.abs(off).max())


def check_independent(log, b, n):
    """Reflectors of one time step must touch disjoint windows (rows and the columns they mix)."""
    by_time = {}
    for t, r0, v, tau in log:
        by_time.setdefault(t, []).append(r0)
    worst = 10 ** 9
    for t, rows in by_time.items():
        rows = sorted(rows)
        for a, c in zip(rows, rows[1:]):
            worst = min(worst, c - a)
    return worst, max(len(v) for v in by_time.values()), len(by_time)


def back(Zt, log, Vh, taus):
    Z = Zt.copy()
    for t, r0, v, tau in sorted(log, key=lambda r: -r[0]):  # reverse time; order inside one time step is free
        rows = slice(r0, r0 + len(v))
        Z[rows] -= tau * np.outer(v, v @ Z[rows])
    for g in range(len(taus) - 1, -1, -1):
        if taus[g] != 0.0:
            v = Vh[g]
            Z -= taus[g] * np.outer(v, v @ Z)
    return Z


def run(A, b, k, name):
    n = A.shape[0]
    band, Vh, taus = stage1(A, b)
    out_of_band = float(np.abs(np.tril(band, -(b + 1))).max()) if n > b + 1 else 0.0
    ev_band = np.linalg.eigvalsh(band)
    ev = np.linalg.eigvalsh(A)
    scale = max(abs(ev).max(), 1e-300)
    d, e, log, off = stage2(band, b)
    T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    w, Zt = np.linalg.eigh(T)
    Z = back(Zt[:, ::-1][:, :k], log, Vh, taus)
    wk = w[::-1][:k]
    res = np.abs(A @ Z - Z * wk).max() / scale
    orth = np.abs(Z.T @ Z - np.eye(k)).max()
    gap, width, steps = check_independent(log, b, n) if log else (0, 0, 0)
    print(f"{name:28s} n={n:4d} b={b} band-ev {np.abs(ev_band - ev).max() / scale:.1e} outside-band {out_of_band:.1e} "
          f"T-ev {np.abs(np.sort(w) - ev).max() / scale:.1e} off-tri {off / scale:.1e} res {res:.1e} orth {orth:.1e} "
          f"reflectors {len(log)} min-row-gap {gap} max-parallel {width} time-steps {steps}")
    assert np.abs(ev_band - ev).max() <= 1e-13 * scale * n and res < 1e-12 * n and orth < 1e-12 * n


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n in (16, 37, 64, 130, 256):
        for b in (2, 4):
            X = rng.standard_normal((2 * n, n))
            run(X.T @ X, b, min(n, 16), "random gram")
            s = np.logspace(0, -6, n)
            X = rng.standard_normal((3 * n, n)) * s
            run(X.T @ X, b, min(n, 16), "graded gram")
    for b in (2, 4):
        u = rng.standard_normal((96, 3))
        run(u @ u.T, b, 8, "rank 3")
        run(np.eye(64), b, 8, "identity")
        run(np.zeros((40, 40)), b, 4, "zero")
        q, _ = np.linalg.qr(rng.standard_normal((80, 80)))
        lam = np.r_[np.ones(10), 0.5 * np.ones(20), np.linspace(0.1, 0.2, 50)]
        run((q * lam) @ q.T, b, 40, "exact multiplicities")
    print("ok")


def stage2_by_time(B, b, lag=3, order=1):
    """The same chase executed time step by time step (what the GPU kernel does); `order` = +1 / -1: the order of the
    independent steps inside one time step.  Must give the same bits as stage2()."""
    n = B.shape[0]
    B = B.copy()
    steps_of = lambda k: -(-(n - 2 - k) // b)  # noqa: E731  rs = k + 1 + m b < n - 1
    t_max = max(lag * k + steps_of(k) for k in range(n - 2)) if n > 2 else 0
    log = []
    for t in range(t_max + 1):
        ks = [k for k in range(n - 2) if 0 <= t - lag * k < steps_of(k)]
        for k in ks[::order]:
            m = t - lag * k
            rs = k + 1 + m * b
            col = k if m == 0 else rs - b
            rows = slice(rs, min(rs + b, n))
            x = B[rows, col].copy()
            if len(x) > 1 and np.any(x[1:] != 0.0):
                v, tau, beta = house(x)
                if tau != 0.0:
                    B[rows, :] -= tau * np.outer(v, v @ B[rows, :])
                    B[:, rows] -= tau * np.outer(B[:, rows] @ v, v)
                log.append((t, rs, v, tau))
    return np.diag(B).copy(), np.diag(B, -1).copy(), log


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    for n, b in ((64, 2), (64, 4), (131, 4), (96, 3)):
        X = rng.standard_normal((2 * n, n)) * np.logspace(0, -4, n)
        band, _, _ = stage1(X.T @ X, b)
        d0, e0, log0, _ = stage2(band, b)
        for order in (1, -1):
            d1, e1, log1 = stage2_by_time(band, b, 3, order)
            same = np.array_equal(d0, d1) and np.array_equal(e0, e1) and len(log0) == len(log1)
            print(f"n={n} b={b} time-major order {order:+d}: identical to the sweep-major chase: {same}")
            assert same
        try:
            d2, e2, _ = stage2_by_time(band, b, 2, -1)
            print("   lag 2 identical:", np.array_equal(d0, d2) and np.array_equal(e0, e2))
        except Exception as exc:
            print("   lag 2 fails:", exc)
