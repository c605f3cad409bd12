"""TEST INFRASTRUCTURE — groundwork for a parallel-in-time Riccati (BASELINE north star: "the serial Riccati recursion run as a
cyclic-reduction / parallel-scan over stages"; DESIGN.md §7 "what comes next").  No product path uses this file.

The backward Riccati sweep is an associative scan over conditional value functions (Särkkä & García-Fernández, "Temporal
parallelization of dynamic programming and linear quadratic control", IEEE TAC 2023).  For the interval i -> j an element
(A, b, C, eta, J) represents

    V_{i->j}(x_i, x_j) = const + 1/2 x_i' J x_i - eta' x_i + max_lam { -1/2 lam' C lam - lam' (x_j - A x_i - b) },

the combination of i -> j with j -> k is

    M  = I + C1 J2,  XA = M^-1 A1,  XC = M^-1 C1,  xb = M^-1 (b1 + C1 eta2),  y = eta2 - J2 b1
    A  = A2 XA            b   = A2 xb + b2            C = A2 XC A2' + C2
    J  = A1' J2 XA + J1   eta = A1' (y - J2 XC y) + eta1          ((I + J2 C1)^-1 = I - J2 M^-1 C1)

and the value function of node k is V_k(x) = 1/2 x' S_k x + s_k' x with S_k = J, s_k = -eta of the suffix k -> N+1.
One stage of the projected QP (x+ = A x + B u + b, cost 1/2 x'Qx + u'Px + 1/2 u'Ru + q'x + r'u) gives, after completing the
square in u,  A - B R^-1 P,  b - B R^-1 r,  C = B R^-1 B',  J = Q - P' R^-1 P,  eta = -(q - P' R^-1 r); the terminal cost is the
element (0, 0, 0, -q_N, Q_N).  A Hillis-Steele suffix scan needs ceil(log2(N + 1)) levels of independent combinations (7 for
N = 100 instead of 100 dependent stages); the gains K_k, k_k then follow for all stages in parallel from S_{k+1}, s_{k+1}.
"""
import numpy as np


def stage_element(A, B, b, Q, P, R, q, r):
    Ri = np.linalg.inv(R)
    return (A - B @ Ri @ P, b - B @ Ri @ r, B @ Ri @ B.T, -(q - P.T @ Ri @ r), Q - P.T @ Ri @ P)


def terminal_element(QN, qN):
    n = QN.shape[0]
    return (np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), -qN, QN.copy())


def combine(e1, e2):
    A1, b1, C1, eta1, J1 = e1
    A2, b2, C2, eta2, J2 = e2
    n = A1.shape[0]
    M = np.eye(n) + C1 @ J2
    X = np.linalg.solve(M, np.concatenate([A1, C1, (b1 + C1 @ eta2)[:, None]], axis=1))   # LU with partial pivoting
    XA, XC, xb = X[:, :n], X[:, n:2 * n], X[:, 2 * n]
    y = eta2 - J2 @ b1
    C = A2 @ XC @ A2.T + C2
    J = A1.T @ J2 @ XA + J1
    return (A2 @ XA, A2 @ xb + b2, 0.5 * (C + C.T), A1.T @ (y - J2 @ (XC @ y)) + eta1, 0.5 * (J + J.T))


def suffix_scan(elements):
    """Hillis-Steele: after the level with stride d, e[k] covers nodes k .. min(k + 2 d, n) - 1.  Returns (suffixes, levels)."""
    e = list(elements)
    n = len(e)
    d, levels = 1, 0
    while d < n:
        e = [combine(e[k], e[k + d]) if k + d < n else e[k] for k in range(n)]   # all combinations of a level are independent
        d *= 2
        levels += 1
    return e, levels


def solve_qp(stages, QN, qN, dx0):
    """stages: list of dicts A, B, b, Q, P, R, q, r.  Returns dx [N+1, nx], u [N, nu], S [N+1], s [N+1], scan levels."""
    N = len(stages)
    elems = [stage_element(**st) for st in stages] + [terminal_element(QN, qN)]
    suf, levels = suffix_scan(elems)
    S = [e[4] for e in suf]
    s = [-e[3] for e in suf]
    dx = [np.asarray(dx0, dtype=float)]
    us = []
    for k, st in enumerate(stages):     # gains of all stages are independent of each other; the roll-out is a chain of mat-vecs
        A, B, b, P, R, r = st["A"], st["B"], st["b"], st["P"], st["R"], st["r"]
        Lam = R + B.T @ S[k + 1] @ B
        K = -np.linalg.solve(Lam, P + B.T @ S[k + 1] @ A)
        kv = -np.linalg.solve(Lam, r + B.T @ (s[k + 1] + S[k + 1] @ b))
        u = K @ dx[-1] + kv
        us.append(u)
        dx.append(A @ dx[-1] + B @ u + b)
    return np.array(dx), np.array(us), S, s, levels
