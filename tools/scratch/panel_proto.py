"""CPU prototype of the panel-blocked tridiagonalisation of csrc/eig_panel.inc (orders above 512), written launch
by launch the way the kernels are: vec(j) finishes column j - 1 (its w from the tile partials of symv(j - 1)) and forms
column j from the matrix of the panel's start and the panel's (V, W) columns; symv(j) multiplies the LOWER TILES of
that matrix with the new reflector; update(p) applies the panel's rank-2NB update to the lower tiles.  Checks T
against numpy's eigenvalues and Q for orthogonality.
usage: python tools/scratch/panel_proto.py [n] [NB] [TB]"""
import sys

import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 32
TB = int(sys.argv[3]) if len(sys.argv) > 3 else 64
TAIL = 128
rng = np.random.default_rng(5)
X = rng.standard_normal((n + 40, n)) * np.logspace(0, -6, n)
G = X.T @ X
A = np.tril(G.copy())  # lower storage; diagonal tiles are kept full below
NTB = (n + TB - 1) // TB
for b in range(NTB):
    s = slice(b * TB, min(n, (b + 1) * TB))
    A[s, s] = G[s, s]
J = max(n - TAIL, 0)  # columns 0 .. J - 1 are reduced here, the rest by the tail kernel
V = np.zeros((n, NB))
W = np.zeros((n, NB))
Vh = np.zeros((n, n))
tau = np.zeros(n)
d = np.zeros(n)
e = np.zeros(n)
u = [np.zeros(n), np.zeros(n)]
sig = [0.0, 0.0]
wtu = [np.zeros(NB), np.zeros(NB)]
vtu = [np.zeros(NB), np.zeros(NB)]
ypart = np.zeros((NTB, n))
s0 = 0.0


def house(alpha, sigma):
    if sigma == 0.0:
        return alpha, 0.0, 0.0
    nrm = np.sqrt(alpha * alpha + sigma)
    beta = -nrm if alpha >= 0 else nrm
    return beta, (beta - alpha) / beta, 1.0 / (alpha - beta)


def panel_of(j):
    return j - j % NB


def vec(j):
    """finish column j - 1, form column j (rows >= j)"""
    wj = 0.0
    i = 0
    if j >= 1:
        jm = j - 1
        i = jm - panel_of(jm)
        x0 = u[jm & 1][j]
        beta, t, scale = house(x0, sig[jm & 1])
        c1 = scale * wtu[jm & 1][:i] + W[j, :i]
        c2 = scale * vtu[jm & 1][:i] + V[j, :i]
        yv = s0 - 2.0 * (c1 @ c2)
        al = 0.5 * t * t * yv
        b0 = j // TB
        y0 = ypart[b0:, :].sum(axis=0)
        v = u[jm & 1] * scale
        v[:j] = 0.0
        v[j] = 1.0
        y = y0 - V[:, :i] @ c1 - W[:, :i] @ c2
        w = t * y - al * v
        w[:j] = 0.0
        V[:, i] = v
        W[:, i] = w
        Vh[jm, :] = v
        d[jm] = u[jm & 1][jm]
        e[jm] = beta
        tau[jm] = t
        i += 1  # columns of the panel now known
    if j >= J:
        return
    # column j from the matrix of the panel's start; at a panel start (i == NB) this is the look-ahead form
    col = np.zeros(n)
    col[j:] = A[j:, j]
    col[j:] -= V[j:, :i] @ W[j, :i] + W[j:, :i] @ V[j, :i] if j >= 1 else 0.0
    u[j & 1] = col
    ut = col.copy()
    ut[: j + 2] = 0.0
    sig[j & 1] = ut @ ut
    if j % NB != 0:  # same panel continues: the corrections of symv(j) need W^T v, V^T v
        wtu[j & 1][:i] = W[:, :i].T @ ut
        vtu[j & 1][:i] = V[:, :i].T @ ut


def symv(j):
    global s0
    beta, t, scale = house(u[j & 1][j + 1], sig[j & 1])
    v = u[j & 1] * scale
    v[: j + 1] = 0.0
    v[j + 1] = 1.0
    b0 = (j + 1) // TB
    ypart[:] = np.nan  # every slot that is read must have been written
    s0 = 0.0
    for R in range(b0, NTB):
        rs = slice(R * TB, min(n, (R + 1) * TB))
        for C in range(b0, R + 1):
            cs = slice(C * TB, min(n, (C + 1) * TB))
            T = A[rs, cs]
            a = T @ v[cs]
            ypart[C, rs] = a
            if R != C:
                ypart[R, cs] = T.T @ v[rs]
                s0 += 2.0 * (v[rs] @ a)
            else:
                s0 += v[rs] @ a


def update(p_next):
    """A[r][c] -= sum_t V[r,t] W[c,t] + W[r,t] V[c,t] on the lower tiles, r, c >= p_next"""
    b0 = p_next // TB
    for R in range(b0, NTB):
        rs = slice(R * TB, min(n, (R + 1) * TB))
        for C in range(b0, R + 1):
            cs = slice(C * TB, min(n, (C + 1) * TB))
            D = V[rs] @ W[cs].T + W[rs] @ V[cs].T
            rr = np.arange(rs.start, rs.stop)[:, None] >= p_next
            cc = np.arange(cs.start, cs.stop)[None, :] >= p_next
            A[rs, cs] -= D * (rr & cc)


ncol = 0
for j in range(J + 1):
    vec(j)
    if j >= 1 and (j % NB == 0 or j == J):
        update(j)
        V[:] = 0.0
        W[:] = 0.0
    if j < J:
        symv(j)
        ncol += 1
# tail: plain dense reduction of the trailing block (full from the lower storage)
m = n - J
S = np.tril(A[J:, J:])
S = S + np.tril(S, -1).T
for b in range(J // TB, NTB):  # diagonal tiles are full: take them as they are
    s = slice(max(b * TB, J) - J, min(n, (b + 1) * TB) - J)
    S[s, s] = A[J:, J:][s, s]
for jj in range(m - 1):
    j = J + jj
    x = S[jj + 1:, jj].copy()
    beta, t, scale = house(x[0], x[1:] @ x[1:])
    v = np.zeros(m)
    v[jj + 1] = 1.0
    v[jj + 2:] = x[1:] * scale
    y = S @ v
    w = t * y - 0.5 * t * t * (y @ v) * v
    d[j] = S[jj, jj]
    e[j] = beta
    tau[j] = t
    Vh[j, J:] = v
    S -= np.outer(v, w) + np.outer(w, v)
d[n - 1] = S[m - 1, m - 1]
Tm = np.diag(d) + np.diag(e[: n - 1], 1) + np.diag(e[: n - 1], -1)
lam = np.linalg.eigvalsh(Tm)[::-1]
ref = np.linalg.eigvalsh(G)[::-1]
Q = np.eye(n)
for j in range(n - 2, -1, -1):
    Q -= tau[j] * np.outer(Vh[j], Vh[j] @ Q)
print(f"n={n} NB={NB} TB={TB}: columns by panels {ncol}, |lam - ref| / lam0 = {np.abs(lam - ref).max() / ref[0]:.2e}, "
      f"|Q^T Q - I| = {np.abs(Q.T @ Q - np.eye(n)).max():.2e}, |Q T Q^T - G| / |G| = "
      f"{np.abs(Q @ Tm @ Q.T - G).max() / np.abs(G).max():.2e}")
