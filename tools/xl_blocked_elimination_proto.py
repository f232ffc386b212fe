"""Development aid (build container, numpy): the blocked elimination of csrc/dgsqp_xl.h (xl_eliminate_blocked) statement by statement
against the column-by-column form it replaces -- M = L~ D L~^T with X = L~^-1 accumulated in place (lower triangle, d on the diagonal).
16 pivots per panel: (a) diagonal block, (b) multipliers of the rows below from their own 16 entries, (c) the panel rows' X by forward
substitution per column (the block's own inverse on the identity), (d) one rank-16 pass over the rows below -- tiles enumerated along the
lower triangle exactly as the kernel does -- plus X[i][panel] = -m_i Xpp.  Garbage above the diagonal must not matter.
usage: python tools/xl_blocked_elimination_proto.py [n ...]"""
import sys
import numpy as np


def column_form(M):
    n = len(M)
    J = np.tril(M).copy()
    for j in range(n - 1):
        d = J[j, j]
        tv, m = np.zeros(n), np.zeros(n)
        tv[:j] = J[j, :j]; tv[j + 1:] = J[j + 1:, j]; m[j + 1:] = J[j + 1:, j] / d
        for i in range(j + 1, n):
            J[i, :i + 1] -= m[i] * tv[:i + 1]          # (column j itself is overwritten next)
            J[i, j] = -m[i]
    return J


def blocked(M, rng):
    n = len(M)
    J = np.tril(M).copy() + np.triu(rng.standard_normal((n, n)), 1)      # garbage above the diagonal
    T = (n + 15) // 16
    for j0 in range(0, n, 16):
        pb, jt = min(16, n - j0), j0 // 16
        a = np.zeros((16, 16))                                             # (a) rows on lanes
        for r in range(pb):
            a[r, :r + 1] = J[j0 + r, j0:j0 + r + 1]
        Lm, dd, dinv = np.zeros((16, 16)), np.zeros(16), np.zeros(16)
        for cc in range(16):
            dcc, live = a[cc, cc], cc < pb
            assert not live or dcc > 0
            inv = 1.0 / dcc if live else 0.0
            m = np.array([a[r, cc] * inv if r > cc else 0.0 for r in range(16)])
            for c2 in range(cc + 1, 16):
                a[c2:, c2] -= m[c2:] * a[c2, cc]
            Lm[:, cc], dd[cc], dinv[cc] = m, (dcc if live else 0.0), inv
            if live:
                J[j0 + cc, j0 + cc] = dcc
        Mm, mrow = np.zeros((16, n)), {}
        for i in range(j0 + 16, n):                                         # (b) thread = row
            ar, m = J[i, j0:j0 + 16].copy(), np.zeros(16)
            for cc in range(16):
                m[cc] = ar[cc] * dinv[cc]
                ar[cc + 1:] -= ar[cc] * Lm[cc + 1:, cc]
            Mm[:, i], mrow[i] = m, m
        Xp = np.zeros((16, 16))
        for k in range(j0 + pb):                                            # (c) thread = column
            kc = k - j0
            x = np.array([(J[j0 + r, k] if r < pb else 0.0) if kc < 0 else float(r == kc) for r in range(16)])
            for r in range(1, 16):
                x[r] -= Lm[r, :r] @ x[:r]
            for r in range(16):
                if r < pb and r > kc:
                    J[j0 + r, k] = x[r]
                if kc >= 0:
                    Xp[r, kc] = x[r]
        if pb < 16 or j0 + 16 >= n:
            break
        for i in range(j0 + 16, n):                                         # (d) the panel's own columns ...
            m = mrow[i]
            for cc in range(16):
                J[i, j0 + cc] = -(m[cc] + m[cc + 1:] @ Xp[cc + 1:, cc])
        base = (jt + 1) * jt // 2                                           # ... and the tiles of the rows below
        Jold = J.copy()
        for t in range(T * (T - 1) // 2 - base):
            g = t + base
            rr = int((1 + np.sqrt(1 + 8 * g)) / 2)
            while rr * (rr - 1) // 2 > g: rr -= 1
            while (rr + 1) * rr // 2 <= g: rr += 1
            cidx = g - rr * (rr - 1) // 2
            ti, tj = rr, (cidx if cidx < jt else cidx + 1)
            for ii in range(16 * ti, min(16 * ti + 16, n)):
                for jj in range(16 * tj, min(16 * tj + 16, n)):
                    b = Jold[j0:j0 + 16, jj] if jj < j0 else Mm[:, jj] * dd
                    J[ii, jj] = Jold[ii, jj] - Mm[:, ii] @ b
    return J


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for n in [int(a) for a in sys.argv[1:]] or [44, 64, 100]:
        A = rng.standard_normal((n, n))
        M = A @ A.T + n * np.eye(n)
        a, b = column_form(M), blocked(M, rng)
        err = np.abs(np.tril(a) - np.tril(b)).max()
        print(f'n = {n}: largest difference in the lower triangle {err:.2e}')
        assert err < 1e-11 * np.abs(a).max()
