#!/usr/bin/env python3
"""NumPy model of the narrow windows' preconditioned eigensolver (csrc/cmf_eigh_pre.h), step for step: Householder
tridiagonalisation -> bisection -> one twisted factorisation per eigenvalue -> reflectors back -> W = L^T U0 -> Newton-Schulz until
the defect measured BEFORE a step is <= 3e-8 -> F = L W'.  Prints what the Jacobi sweeps would find (largest cosine between
columns of F) and how far F F^T is from R, on flightline-like correlation matrices (a noise-floor cluster + a few signal
directions; cmf/robust_mf.py:92-136 is what consumes the eigenpairs).  CPU only: the development check of the numerics."""
import sys
import numpy as np


def tridiag(A):
    A = A.copy(); n = A.shape[0]
    V = np.zeros((n, n)); tau = np.zeros(n); d = np.zeros(n); e = np.zeros(n)
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        alpha = x[0]; s = float(x[1:] @ x[1:])
        if s == 0.0:
            tau[k] = 0.0; e[k] = alpha; v = np.zeros_like(x); v[0] = 1.0
        else:
            beta = -np.copysign(np.sqrt(alpha * alpha + s), alpha)
            tau[k] = (beta - alpha) / beta
            v = x / (alpha - beta); v[0] = 1.0
            e[k] = beta
        V[k + 1:, k] = v
        T = A[k + 1:, k + 1:]
        pv = tau[k] * (T @ v)
        K = 0.5 * tau[k] * float(v @ pv)
        w = pv - K * v
        T -= np.outer(v, w) + np.outer(w, v)
        d[k] = A[k, k]
    d[n - 2] = A[n - 2, n - 2]; d[n - 1] = A[n - 1, n - 1]; e[n - 2] = A[n - 1, n - 2]
    return d, e[:n - 1], V, tau


def sturm(d, e2, x, tiny):
    q = d[0] - x
    if q == 0.0: q = -tiny
    c = int(q < 0)
    for i in range(1, len(d)):
        q = d[i] - x - e2[i - 1] / q
        if q == 0.0: q = -tiny
        c += int(q < 0)
    return c


def bisect_all(d, e, rounds=28):
    n = len(d); e2 = e * e
    rad = np.zeros(n); rad[:-1] += np.abs(e); rad[1:] += np.abs(e)
    lo0, hi0 = (d - rad).min(), (d + rad).max()
    w = hi0 - lo0
    lo0 -= 1e-3 * w + 1e-300; hi0 += 1e-3 * w + 1e-300
    tnorm = max(abs(lo0), abs(hi0)); tiny = 2.2e-16 * tnorm * 1e-3 + 1e-300
    lam = np.zeros(n)
    for k in range(n):
        lo, hi = lo0, hi0
        for _ in range(rounds):
            xs = [lo + (j + 1) * (hi - lo) / 5.0 for j in range(4)]
            cs = [sturm(d, e2, x, tiny) for x in xs]
            nlo, nhi = lo, hi
            for j in range(4):
                if cs[j] > k:
                    nhi = xs[j]; break
                nlo = xs[j]
            lo, hi = nlo, nhi
        lam[k] = 0.5 * (lo + hi)
    return lam, tiny


def twisted(d, e, lam, tiny):
    n = len(d); e2 = e * e
    Dp = np.zeros(n); Dm = np.zeros(n)
    Dp[0] = d[0] - lam or tiny
    for i in range(n - 1):
        Dp[i + 1] = (d[i + 1] - lam) - e2[i] / Dp[i]
        if Dp[i + 1] == 0.0: Dp[i + 1] = tiny
    Dm[n - 1] = d[n - 1] - lam or tiny
    for i in range(n - 2, -1, -1):
        Dm[i] = (d[i] - lam) - e2[i] / Dm[i + 1]
        if Dm[i] == 0.0: Dm[i] = tiny
    g = np.abs(Dp + Dm - (d - lam))
    r = int(np.argmin(g))
    z = np.zeros(n); z[r] = 1.0
    for i in range(r - 1, -1, -1):
        z[i] = -(e[i] / Dp[i]) * z[i + 1]
    for i in range(r, n - 1):
        z[i + 1] = -(e[i] / Dm[i + 1]) * z[i]
    return z / np.linalg.norm(z)


def precondition(R, verbose=True):
    n = R.shape[0]
    d, e, V, tau = tridiag(R)
    lam, tiny = bisect_all(d, e)
    if not np.all(lam > 0): return None
    Z = np.stack([twisted(d, e, l, tiny) for l in lam], 1)
    U = Z.copy()
    for k in range(n - 3, -1, -1):
        v = V[:, k]
        U -= tau[k] * np.outer(v, v @ U)
    L = np.linalg.cholesky(R)
    W = L.T @ U
    W /= np.linalg.norm(W, axis=0, keepdims=True)
    steps = 0
    while True:
        G = W.T @ W
        defect = np.abs(G - np.eye(n)).max()
        if defect > 0.3: return None
        W = W @ (1.5 * np.eye(n) - 0.5 * G); steps += 1
        if defect <= 3e-8 or steps == 4: break
    F = L @ W
    nr = np.linalg.norm(F, axis=0)
    C = (F.T @ F) / np.outer(nr, nr); np.fill_diagonal(C, 0)
    if verbose:
        print("  tri eig rel err %.1e | NS steps %d (first defect %.1e) | max cosine of F %.1e | |FF^T - R| %.1e | lam rel err %.1e"
              % (np.abs(np.sort(lam) - np.linalg.eigvalsh(R)).max() / lam.max(), steps, defect0(L, U), np.abs(C).max(),
                 np.abs(F @ F.T - R).max(), np.abs(np.sort(nr ** 2) - np.linalg.eigvalsh(R)).max() / (nr ** 2).max()))
    return F


def defect0(L, U):
    W = L.T @ U; W /= np.linalg.norm(W, axis=0, keepdims=True)
    return np.abs(W.T @ W - np.eye(W.shape[0])).max()


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    for p, lines, decades in ((72, 20000, 2.5), (72, 400, 2.5), (83, 20000, 4.0), (72, 20000, 7.0), (72, 90, 2.0)):
        base = 5 * np.exp(-3 * np.arange(p) / (p - 1)) + 0.2
        Lm = rng.normal(size=(5, p)) * 0.1 * base
        x = base + rng.normal(size=(lines, 5)) @ Lm + rng.normal(size=(lines, p)) * (10.0 ** (-decades / 2.5 * 0.8)) * base
        S = np.cov(x.T)
        dd = np.sqrt(np.diag(S)); R = S / np.outer(dd, dd)
        print("p %d lines %d: cond %.1e" % (p, lines, np.linalg.cond(R)))
        precondition(R)
