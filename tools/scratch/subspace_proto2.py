"""CPU prototype of the GPU algorithm: block subspace iteration with a degree-`deg` Chebyshev filter,
single-pass scaled Cholesky-QR between filters (no Rayleigh-Ritz inside the loop), Ritz step only when the
R diagonal has settled; residual check; returns status for the Jacobi fallback."""
import sys
import numpy as np

def chol_qr(Z, passes=1):
    """Z -> Q with Q^T Q ~ I via diagonally scaled Cholesky of Z^T Z; returns (Q, diag of total R) or None."""
    rdiag = np.ones(Z.shape[1])
    for _ in range(passes):
        S = Z.T @ Z
        d = np.sqrt(np.diag(S))
        if not np.all(d > 0) or not np.all(np.isfinite(d)): return None
        Ss = S / d[:, None] / d[None, :]
        try:
            Lc = np.linalg.cholesky(Ss)
        except np.linalg.LinAlgError:
            return None
        if np.min(np.diag(Lc)) < 1e-7: return None        # numerically rank deficient block
        Rinv = np.linalg.inv(Lc.T) / d[:, None]          # (D R')^-1 = R'^-1 D^-1 -> columns scaled... see below
        # Z = Q R with R = R' D  =>  Q = Z D^-1 R'^-1
        Q = (Z / d[None, :]) @ np.linalg.inv(Lc.T)
        rdiag = rdiag * np.diag(Lc) * d
        Z = Q
    return Z, rdiag

def topk(G, k, b=128, deg=2, tol=1e-10, max_matvecs=200, seed=0, verbose=False):
    n = G.shape[0]
    rng = np.random.default_rng(seed)
    r = chol_qr(rng.standard_normal((n, b)), 2)
    Q = r[0]
    mv = 0
    prev = None
    lam1 = None
    a = None
    while mv < max_matvecs:
        Z = G @ Q; mv += 1
        if a is not None and deg >= 2:
            c = a / 2.0; e = a / 2.0
            Y0 = Q; Y1 = (Z - c * Q) / e
            for j in range(2, deg + 1):
                Y2 = 2.0 * ((G @ Y1) - c * Y1) / e - Y0; mv += 1
                Y0, Y1 = Y1, Y2
            Z = Y1
        r = chol_qr(Z, 1)
        if r is None: return None, None, mv, "breakdown"
        Q, rd = r
        if a is None or deg < 2:
            est = rd           # R diagonal of G Q: eigenvalue estimates
            # next filter: damp [0, a] with a = smallest estimate
            newa = est.min()
            settled = prev is not None and np.abs(np.sort(est)[::-1][:k] - prev).max() <= 1e-12 * est.max()
            prev = np.sort(est)[::-1][:k]
        else:
            settled = False
            # every few filtered steps do an unfiltered one to refresh estimates
        a_next = None
        if a is None:
            a = newa
        else:
            a = None  # alternate: filtered step, then a plain step (refresh estimates + convergence test)
            continue
        if settled or mv >= max_matvecs - 2:
            break
    # final Rayleigh-Ritz on an accurately orthonormal basis
    r = chol_qr(Q, 2)
    if r is None: return None, None, mv, "breakdown"
    Q = r[0]
    Z = G @ Q; mv += 1
    H = Q.T @ Z; H = 0.5 * (H + H.T)
    th, Y = np.linalg.eigh(H); th = th[::-1]; Y = Y[:, ::-1]
    X = Q @ Y; GX = Z @ Y
    res = np.linalg.norm(GX[:, :k] - X[:, :k] * th[:k], axis=0).max() / th[0]
    return th, X, mv, ("ok" if res < tol else f"residual {res:.1e}")

if __name__ == "__main__":
    d = np.load(sys.argv[1]); k = 64
    for name in d.files:
        G = d[name]
        w, V = np.linalg.eigh(G); w = w[::-1]; V = V[:, ::-1]
        for deg in (1, 2, 3):
            th, X, mv, status = topk(G, k, 128, deg)
            if th is None: print(name, "deg", deg, status, mv); continue
            Vk = V[:, :k]; Xk = X[:, :k]; D = Xk @ Xk.T - Vk @ Vk.T
            rel = np.sqrt(abs(np.trace(D @ G @ D))) / np.sqrt(np.trace(G))
            print(f"{name} deg={deg}: matvecs={mv} status={status} rel_recon_diff={rel:.1e} eigval relerr={np.abs(th[:k]-w[:k]).max()/w[0]:.1e}")
