"""CPU prototype: Chebyshev-filtered block subspace iteration + Rayleigh-Ritz for the top-k eigenpairs
of a symmetric PSD matrix, with CholQR2 orthonormalisation (the operations the GPU version would use)."""
import sys
import numpy as np

def cholqr2(Z):
    for _ in range(2):
        S = Z.T @ Z
        # scale-aware jitter-free Cholesky; raises if not PD
        R = np.linalg.cholesky(S).T
        Z = np.linalg.solve(R.T, Z.T).T   # Z R^{-1}
    return Z

def topk(G, k, b, deg, tol, max_outer=60, seed=0, verbose=False):
    n = G.shape[0]
    rng = np.random.default_rng(seed)
    Q = cholqr2(rng.standard_normal((n, b)))
    lam1 = None
    matvecs = 0
    hist = []
    for outer in range(max_outer):
        # Rayleigh-Ritz
        Z = G @ Q; matvecs += 1
        H = Q.T @ Z
        th, Y = np.linalg.eigh(H); th = th[::-1]; Y = Y[:, ::-1]
        X = Q @ Y; GX = Z @ Y
        res = np.linalg.norm(GX[:, :k] - X[:, :k] * th[:k], axis=0)
        r = res.max() / th[0]
        hist.append((matvecs, r))
        if verbose: print(f"  outer {outer} matvecs {matvecs} max residual/lam1 {r:.2e}  th[k-1]/th[0]={th[k-1]/th[0]:.2e} th[b-1]/th[k-1]={th[b-1]/th[k-1]:.3f}")
        if r < tol: return th, X, matvecs, hist
        # Chebyshev filter damping [0, a], a = smallest Ritz value
        a = th[-1]
        c = a / 2.0; e = a / 2.0
        # three-term recurrence on the shifted matrix (G - c I)/e
        Y0 = X
        Y1 = (GX - c * X) / e           # uses the product already computed
        for j in range(2, deg + 1):
            Y2 = 2.0 * ((G @ Y1) - c * Y1) / e - Y0; matvecs += 1
            Y0, Y1 = Y1, Y2
        Q = cholqr2(Y1)
    return th, X, matvecs, hist

if __name__ == "__main__":
    d = np.load(sys.argv[1])
    k = 64
    for name in d.files:
        G = d[name]
        w, V = np.linalg.eigh(G); w = w[::-1]; V = V[:, ::-1]
        s = np.sqrt(np.maximum(w, 0))
        print(name, f"s64/s0={s[63]/s[0]:.2e} s65/s64={s[64]/s[63]:.4f} s128/s64={s[127]/s[63]:.3f} s192/s64={s[191]/s[63]:.3f}")
        for b in (128, 192):
            for deg in (1, 2, 4, 6):
                try:
                    th, X, mv, hist = topk(G, k, b, deg, 1e-11)
                except np.linalg.LinAlgError as ex:
                    print(f"   b={b} deg={deg}: cholesky failed"); continue
                Vk = V[:, :k]; Xk = X[:, :k]
                D = Xk @ Xk.T - Vk @ Vk.T
                rel = np.sqrt(abs(np.trace(D @ G @ D))) / np.sqrt(np.trace(G))
                print(f"   b={b} deg={deg}: matvecs={mv} outer={len(hist)} final res={hist[-1][1]:.1e} rel_recon_diff={rel:.1e} eigval relerr={np.abs(th[:k]-w[:k]).max()/w[0]:.1e}")
