"""Numerical prototype of the speed-bias-first elimination used by k_solve (not product code).
H over [poses 66 | SB 11x9]: SB-SB block tridiagonal (IMU factors couple consecutive frames), SB-pose banded + one dense row block (prior, SB0)."""
import numpy as np
rng = np.random.default_rng(0)
NP, NS = 66, 99
J = []
# visual-like dense pose terms
A = rng.standard_normal((300, NP)); Hpp = A.T @ A
H = np.zeros((165, 165)); H[:NP, :NP] = Hpp
def sidx(a): return slice(NP + 9 * a, NP + 9 * a + 9)
def pidx(a): return slice(6 * a, 6 * a + 6)
for k in range(10):          # IMU factor k couples pose k, sb k, pose k+1, sb k+1
    Jk = rng.standard_normal((15, 30))
    idx = np.r_[np.arange(6 * k, 6 * k + 6), np.arange(NP + 9 * k, NP + 9 * k + 9), np.arange(6 * k + 6, 6 * k + 12), np.arange(NP + 9 * k + 9, NP + 9 * k + 18)]
    H[np.ix_(idx, idx)] += Jk.T @ Jk
# prior: poses 0..9 + SB0
Jp = rng.standard_normal((69, 69)); idx = np.r_[np.arange(60), np.arange(NP, NP + 9)]
H[np.ix_(idx, idx)] += Jp.T @ Jp
H += 1e-3 * np.eye(165)
g = rng.standard_normal(165)
y_ref = np.linalg.solve(H, g)

Hss = H[NP:, NP:]; Hsp = H[NP:, :NP]
D = [Hss[9 * a:9 * a + 9, 9 * a:9 * a + 9].copy() for a in range(11)]
E = [Hss[9 * (a + 1):9 * (a + 1) + 9, 9 * a:9 * a + 9].copy() for a in range(10)]   # E_a = H(s_{a+1}, s_a)
assert np.allclose(Hss, sum(np.kron(np.eye(11)[:, [a]] @ np.eye(11)[[a], :], D[a]) for a in range(11)) +
                   sum(np.kron(np.eye(11)[:, [a + 1]] @ np.eye(11)[[a], :], E[a]) + np.kron(np.eye(11)[:, [a]] @ np.eye(11)[[a + 1], :], E[a].T) for a in range(10)))
L = [None] * 11; Bm = [None] * 12
Dp = [d.copy() for d in D]
for a in range(10, -1, -1):
    L[a] = np.linalg.cholesky(Dp[a])
    if a > 0:
        Bm[a] = np.linalg.solve(L[a], E[a - 1])       # B_a = L_a^-1 E_{a-1}   (rows s_a, cols s_{a-1})
        Dp[a - 1] -= Bm[a].T @ Bm[a]
# Y = L^-1 [Hsp | g_s]
R = np.hstack([Hsp, g[NP:, None]])
Y = [None] * 11
for a in range(10, -1, -1):
    t = R[9 * a:9 * a + 9].copy()
    if a < 10: t -= Bm[a + 1].T @ Y[a + 1]
    Y[a] = np.linalg.solve(L[a], t)
Yall = np.vstack(Y)
S = H[:NP, :NP] - Yall[:, :NP].T @ Yall[:, :NP]
rhs = g[:NP] - Yall[:, :NP].T @ Yall[:, NP]
yp = np.linalg.solve(S, rhs)
# back-substitute: Hss ys = g_s - Hsp yp
r = g[NP:] - Hsp @ yp
u = [None] * 11
for a in range(10, -1, -1):
    t = r[9 * a:9 * a + 9].copy()
    if a < 10: t -= Bm[a + 1].T @ u[a + 1]
    u[a] = np.linalg.solve(L[a], t)
x = [None] * 11
for a in range(11):
    t = u[a].copy()
    if a > 0: t -= Bm[a] @ x[a - 1]
    x[a] = np.linalg.solve(L[a].T, t)
y = np.r_[yp, np.concatenate(x)]
print("max rel err", np.abs(y - y_ref).max() / np.abs(y_ref).max())
# staircase structure of Y
nz = [np.nonzero(np.abs(Y[a][:, :NP]).max(0) > 1e-14)[0].min() for a in range(11)]
print("first nonzero pose column of Y_a:", nz)

# ---- product form (k_solve_sb v2): chain blocks a = 1..10 only (SpeedBias[0] belongs to the dense part), M_a = L_a^-1, N_a = M_a B_(a+1)^T ----
ND = 75
Hd = H[:ND, :ND]; Hcd = H[ND:, :ND]; Hcc = H[ND:, ND:]
Dc = [Hcc[9 * k:9 * k + 9, 9 * k:9 * k + 9].copy() for k in range(10)]          # index k = a - 1
Ec = [Hcc[9 * (k + 1):9 * (k + 1) + 9, 9 * k:9 * k + 9].copy() for k in range(9)]
Lc = [None] * 10; Bc = [None] * 10
for k in range(9, -1, -1):
    Lc[k] = np.linalg.cholesky(Dc[k])
    if k > 0:
        Bc[k] = np.linalg.solve(Lc[k], Ec[k - 1]); Dc[k - 1] -= Bc[k].T @ Bc[k]
M = [np.linalg.inv(l) for l in Lc]
N = [M[k] @ Bc[k + 1].T for k in range(9)]                                      # N_a, a = k + 1 = 1..9
R = np.hstack([Hcd, g[ND:, None]])
Y = [None] * 10
for k in range(9, -1, -1):
    Y[k] = M[k] @ R[9 * k:9 * k + 9] - (N[k] @ Y[k + 1] if k < 9 else 0)
Yall = np.vstack(Y)
S = Hd - Yall[:, :ND].T @ Yall[:, :ND]; rhs = g[:ND] - Yall[:, :ND].T @ Yall[:, ND]
yd = np.linalg.solve(S, rhs)
r = g[ND:] - Hcd @ yd
c = [M[k] @ r[9 * k:9 * k + 9] for k in range(10)]
u = [None] * 10
for k in range(9, -1, -1):
    u[k] = c[k] - (N[k] @ u[k + 1] if k < 9 else 0)
wv = [None] * 10; wv[0] = u[0]
for k in range(9):
    wv[k + 1] = u[k + 1] - N[k].T @ wv[k]
x = [M[k].T @ wv[k] for k in range(10)]
y2 = np.r_[yd, np.concatenate(x)]
print("product form: max rel err", np.abs(y2 - y_ref).max() / np.abs(y_ref).max())
