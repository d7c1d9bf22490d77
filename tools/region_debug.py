"""Block map of |L_region - L_lapack| (max per 64 x 64 block) for lmm_dev_potrf on a random SPD matrix: python tools/region_debug.py n nrider"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
lmm_amd.init(0); lib = lmm_amd.load()
n, nrider = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(n + nrider)
NC = (n + 63) // 64 * 64; NR = (NC + nrider + 63) // 64 * 64; ld = NR + 16
G = rng.standard_normal((n, n + 8)); K = G @ G.T / (n + 8) + 0.5 * np.eye(n)
R = rng.standard_normal((NR - NC, NC))
full = np.zeros((NR, NC)); full[:n, :n] = np.tril(K); full[n:NC, n:] = np.eye(NC - n); full[NC:, :] = R
for trial in range(3):
    A = torch.zeros((NC, ld), dtype=torch.float64, device="cuda"); A[:, :NR] = torch.from_numpy(np.ascontiguousarray(full.T)).cuda()
    W = torch.zeros((NC // 64, 64, 64), dtype=torch.float64, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.lmm_dev_potrf(C.c_void_p(A.data_ptr()), NR, NC, ld, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr()))
    got = A.cpu().numpy().T[:NR]
    Kp = np.eye(NC); Kp[:n, :n] = K; Lref = np.linalg.cholesky(Kp)
    import scipy.linalg as sla
    ref = np.zeros((NR, NC)); ref[:NC] = np.tril(Lref); ref[NC:] = sla.solve_triangular(Lref, R.T, lower=True).T
    err = np.abs(np.tril(got[:NC]) - ref[:NC]); errb = np.abs(got[NC:] - ref[NC:])
    E = np.vstack([err, errb])
    nbr, nbc = NR // 64, NC // 64
    print(f"trial {trial} rc={rc} info={int(info.item())} max err {E.max():.3e}")
    for i in range(nbr):
        print("".join("." if E[64*i:64*i+64, 64*j:64*j+64].max() < 1e-10 else "X" for j in range(min(i + 1, nbc))))
