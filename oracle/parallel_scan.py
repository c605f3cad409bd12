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


def solve_qp(stages, QN, qN, dx0, prepended=False):
    """stages: list of dicts A, B, b, Q, P, R, q, r.  Returns dx [N+1, nx], u [N, nu], S [N+1], s [N+1], scan levels."""
    N = len(stages)
    # prepended: the stage elements the device forms since round 5 (hsqp_scan.h::scan_init_node): the stage prepended to the empty interval —
    # R = L L', Z = L^-1 P, W = B L^-T, A - W Z, C = W W', J = Q - Z'Z — instead of products with the explicit R^-1 (cond(R) ~ 1e7 on the whole-body problem)
    n = QN.shape[0]
    elems = [(prepend_stage(st, identity_element(n)) if prepended else stage_element(**st)) for st in stages] + [terminal_element(QN, qN)]
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


# ------------------------------------------------------------------------------------------------------------------------------
# Two-level (segmented) sweep — the strong-scaling form for 2 < B <= 128 instances per GPU: the horizon is cut into
# P segments; (1) every segment's element by PREPENDING its stages one at a time, (2) a suffix scan over the P segment elements and
# the terminal element gives the value function at every segment boundary, (3) an ordinary Riccati recursion per segment from its
# boundary value function gives the gains, (4) the usual roll-out.  (1) and (3) are parallel over segments.
#
# Prepending stage k (x+ = A x + B u + b, cost Q, P, R, q, r) to the element (A2, b2, C2, eta2, J2) of k+1 .. end: the stage's
# element has C1 = B R^-1 B' of rank nu, so M^-1 = (I + C1 J2)^-1 follows from the Riccati gain solve with S = J2, s = -eta2:
#     Lam = R + B' S B,  G = P + B' S A,  g = r + B' (S b + s),  K = -Lam^-1 G,  k = -Lam^-1 g
#     M^-1 A1 = A + B K = Acl,   M^-1 (b1 + C1 eta2) = b + B k = bcl,   M^-1 C1 = B Lam^-1 B'
#     J <- Q + A' S A - G' Lam^-1 G,   eta <- -(q + A' (S b + s) - G' Lam^-1 g)            (the Riccati step itself)
#     A <- A2 Acl,   b <- A2 bcl + b2,   C <- A2 B Lam^-1 B' A2' + C2
def prepend_stage(st, e2):
    A2, b2, C2, eta2, J2 = e2
    A, B, b, Q, P, R, q, r = (st[k] for k in ("A", "B", "b", "Q", "P", "R", "q", "r"))
    S, s = J2, -eta2
    Lam = R + B.T @ S @ B
    G = P + B.T @ S @ A
    sb = S @ b + s
    g = r + B.T @ sb
    L = np.linalg.cholesky(Lam)
    Z = np.linalg.solve(L, G)
    z = np.linalg.solve(L, g)
    K = -np.linalg.solve(L.T, Z)
    kv = -np.linalg.solve(L.T, z)
    J = Q + A.T @ S @ A - Z.T @ Z
    eta = -(q + A.T @ sb - Z.T @ z)
    Acl, bcl = A + B @ K, b + B @ kv
    W = A2 @ np.linalg.solve(L, B.T).T          # A2 B L^-T
    return (A2 @ Acl, A2 @ bcl + b2, W @ W.T + C2, eta, 0.5 * (J + J.T))


def identity_element(n):
    return (np.eye(n), np.zeros(n), np.zeros((n, n)), np.zeros(n), np.zeros((n, n)))


def riccati_segment(stages, S, s):
    """Ordinary backward recursion over `stages` from the value function (S, s) at their end: gains (K, k) per stage and (S, s) at the start."""
    gains = []
    for st in reversed(stages):
        A, B, b, Q, P, R, q, r = (st[k] for k in ("A", "B", "b", "Q", "P", "R", "q", "r"))
        Lam = R + B.T @ S @ B
        G = P + B.T @ S @ A
        sb = S @ b + s
        g = r + B.T @ sb
        K = -np.linalg.solve(Lam, G)
        kv = -np.linalg.solve(Lam, g)
        S, s = Q + A.T @ S @ A + G.T @ K, q + A.T @ sb + G.T @ kv
        S = 0.5 * (S + S.T)
        gains.append((K, kv))
    return gains[::-1], S, s


# Guess shift (round 5).  A quadratic 1/2 x' G x at a boundary node is the element c(G) = (I, 0, 0, 0, G); c(-G) o c(G) is the identity, so the
# chain of elements e_0 o e_1 o .. equals  e_0 o c(G_1)  o  c(-G_1) o e_1 o c(G_2)  o  c(-G_2) o e_2 ..: every segment gets the guess of the value
# function at its END as a terminal cost (its recursion then starts from S = G_{p+1} instead of S = 0: Lam = R + B' G B is as well conditioned
# as in the serial recursion, where R alone has condition 1e7 on the whole-body problem) and hands the guess at its START back (c(-G) o e: J - G,
# nothing else changes, exactly).  The scan then carries J_p = S_p - G_p: the CORRECTION of the guess.  With the boundary value functions of the
# previous SQP iteration as guesses |C J| << 1 and M = I + C1 J2 is near the identity; with any guess the suffixes are the same in exact
# arithmetic.  Measured (whole-body, N = 100, walk): step error against the serial recursion 8e-9 without guesses, 7e-10 with G = 0 but the
# elements formed this way, 1e-9 with the value functions of a differently perturbed problem, 3e-11 with guesses 0.1 % off.
def shifted_segment_element(stages, G_end, G_start):
    n = G_end.shape[0]
    e = (np.eye(n), np.zeros(n), np.zeros((n, n)), np.zeros(n), G_end)
    for st in reversed(stages):
        e = prepend_stage(st, e)
    return (e[0], e[1], e[2], e[3], e[4] - G_start)


def solve_qp_segmented(stages, QN, qN, dx0, n_segments, guesses=None):
    """Returns dx, u, the boundary indices, the boundary value functions and the number of scan levels over the segments.
    guesses (optional): n_segments + 1 symmetric matrices, guesses of the value function's Hessian at the segment boundaries."""
    N, n = len(stages), QN.shape[0]
    bounds = [round(p * N / n_segments) for p in range(n_segments + 1)]
    G = [np.zeros((n, n))] * (n_segments + 1) if guesses is None else list(guesses)
    # (1) segment elements, independent of each other
    elems = []
    for p in range(n_segments):
        if guesses is None:
            e = identity_element(n)
            for k in range(bounds[p + 1] - 1, bounds[p] - 1, -1):
                e = prepend_stage(stages[k], e)
        else:
            e = shifted_segment_element(stages[bounds[p]:bounds[p + 1]], G[p + 1], G[p])
        elems.append(e)
    # (2) suffix scan over the segment elements + the terminal element: value function at every boundary
    te = terminal_element(QN, qN)
    suf, levels = suffix_scan(elems + [(te[0], te[1], te[2], te[3], te[4] - G[n_segments])])
    suf = [(e[0], e[1], e[2], e[3], e[4] + G[p]) for p, e in enumerate(suf)]
    # (3) gains per segment from the value function at its END, (4) roll-out
    gains = []
    for p in range(n_segments):
        g, _, _ = riccati_segment(stages[bounds[p]:bounds[p + 1]], suf[p + 1][4], -suf[p + 1][3])
        gains += g
    dx, us = [np.asarray(dx0, dtype=float)], []
    for st, (K, kv) in zip(stages, gains):
        u = K @ dx[-1] + kv
        us.append(u)
        dx.append(st["A"] @ dx[-1] + st["B"] @ u + st["b"])
    return np.array(dx), np.array(us), bounds, [(e[4], -e[3]) for e in suf], levels
