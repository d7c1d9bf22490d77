#!/usr/bin/env python3
"""bench.py -- logpdf evaluations/s of the ILMM/OILMM hot path on N MI355X (BASELINE.json metric).

A "step" is ONE logpdf evaluation of the workload (all n x p observations) with x and y already resident in
HBM.  Default workload = BASELINE.json configs[2] (the configuration north_star quotes its scaling target on):
OILMM, Orthogonal(U,S) 64x32, 32 Matern52 latents, n = 16384, Float64.  It fits one GPU, so the same problem is
run at every N with the 32 latents sharded over the ranks (strong scaling; one scalar all-reduce per step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c1dense|c0|small|notebook|c3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the field definitions).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (m, p, n, kernel, orthogonal, BASELINE.json config it is)
    "c2": (32, 64, 16384, "matern52", True, "configs[2]: OILMM, Orthogonal(U,S) 64x32, 32 Matern52 latents, n=16384, f64"),
    "c1": (8, 16, 4096, "se", False, "configs[1]: ILMM, dense H 16x8, 8 SEKernel latents, n=4096, f64 (identical kernels: "
                                     "decoupled shortcut, m independent n x n factorisations)"),
    "c1dense": (8, 16, 4096, "se", False, "configs[1]: ILMM, dense H 16x8, 8 SEKernel latents, n=4096, f64 (the reference's "
                                          "single (mn)x(mn) factorisation, shortcut disabled)"),
    "c0": (3, 5, 200, "se", True, "configs[0]: OILMM, 3 SEKernel latents, p=5, n=200, f64"),
    "small": (8, 16, 2048, "matern52", True, "reduced smoke workload (NOT a BASELINE config)"),
    # secondary metric: one step = posterior(fx, y) + marginals at n* = n test points, latents sharded, ONE
    # all-reduce of the p x n* partial means / variances per step.  `--proj bf16` runs the configuration as named:
    # the H unprojection of the marginals on v_mfma_f32_16x16x32_bf16 (lmm_set_projection_dtype; latent marginals stay f64).
    "c3": (64, 128, 8192, "matern52", True, "configs[3]: OILMM posterior predictive, H 128x64, n_train = n_test = 8192 "
                                            "(step = posterior + mean_and_var at x*; --proj bf16 = the bf16 MFMA covariance projection)"),
    # the reference notebook's timing shape (examples/oilmm_and_ilmm.ipynb:124-129, 226-235): the only published number
    "notebook": (20, 600, 552, "matern52", True, "reference notebook: OILMM logpdf, p=600, m=20, n=552 Matern52, sigma2=1e-6, f64"),
    # secondary metric: BASELINE configs[4] -- one step = ONE prior sample rand(rng, fx) of the whole model: a Cholesky of every
    # latent Gram + the sampling transform L z + the H mix, latents sharded, ONE all-reduce of the n x p partial sample per step.
    # Run with --dtype f32 for the configuration as named (latent jitter 1e-4 x variance: SURVEY.md section 7 "jitter hazards").
    "c4": (128, 256, 32768, "matern52", True, "configs[4]: OILMM rand, 256 outputs, 128 Matern52 latents, n=32768 "
                                              "(step = one prior sample: per-latent Cholesky + L z + mix)"),
}
NOTEBOOK_PUBLISHED_EVALS_PER_S = 1.0 / 0.172541   # median 172.541 ms, unstated CPU, Julia 1.6.1
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (155 measured)
FP64_MFMA_PEAK_TFLOPS = 78.6      # AMD MI355X datasheet FP64 matrix (= vector) peak; the local guide lists no FP64 MFMA
                                  # rate.  Measured here: v_mfma_f64_16x16x4_f64 with VGPR accumulators issues at 77.7 TFLOP/s
                                  # (tools/mfma_probe4, profiles/r02) = 98.9 % of it, so 78.6 is the denominator.
HBM_PEAK_GBS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def build_model(lmm, P):
    K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
    fs = lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in P["gps"]])
    H = lmm.Orthogonal(P["U"], P["S"]) if "U" in P else P["H"]
    return fs, H


def cpu_baseline(P, orthogonal, budget_latents=2):
    """The oracle (a NumPy/SciPy port of the reference algorithm: per-latent Gram -> LAPACK dpotrf -> dtrtrs) timed
    on the host cores on a bounded sample of the SAME workload: `budget_latents` of the m latents at full n,
    extrapolated linearly to m latents (latents are independent and equally sized)."""
    from oracle import lmm_oracle as O
    m, n = P["m"], P["n"]
    cores = os.cpu_count() or 1
    if not orthogonal:
        # dense ILMM: (mn)^3/3 does not subsample by latents; time a reduced n and scale by the cubic flop count
        ns = max(64, n // 2)      # (m n/2)^3/3 flops: seconds of LAPACK at C1, large enough that fixed overheads do not dominate
        from lmm_amd.workloads import synthetic_problem
        Ps = synthetic_problem(m, P["p"], ns, P["gps"][0]["kind"], False, P["s2"], seed=0)
        t0 = time.perf_counter(); O.ilmm_logpdf(Ps["gps"], Ps["H"], Ps["x"], Ps["s2"], Ps["y"]); dt = time.perf_counter() - t0
        est = dt * (n / ns) ** 3
        return {"value": 1.0 / est, "unit": "evals/s", "cores": cores, "kind": "port",
                "sample": f"oracle ilmm_logpdf at n={ns} ({dt:.2f} s), scaled by (n/{ns})^3 to n={n}"}
    if m * n ** 3 / 3.0 < 2e10:
        # small shapes (BASELINE configs[0], the reference notebook's shape): the WHOLE evaluation as the reference runs it
        # (src/oilmm.jl:79-93: project, per-latent logpdf, regulariser), SURVEY.md 8d protocol: 2 warm-ups, median of >= 10 runs
        def once():
            t = time.perf_counter()
            O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], P["s2"], P["y"])
            return time.perf_counter() - t
        for _ in range(2):
            once()
        runs = sorted(once() for _ in range(15))
        med = runs[len(runs) // 2]
        return {"value": 1.0 / med, "unit": "evals/s", "cores": cores, "kind": "port",
                "sample": f"oracle oilmm_logpdf (NumPy projection + Gram, LAPACK potrf/trtrs per latent, regulariser), all {m} latents at n={n}: "
                          f"median of {len(runs)} runs after 2 warm-ups = {med * 1e6:.1f} us (min {runs[0] * 1e6:.1f}, max {runs[-1] * 1e6:.1f})"}
    k = min(budget_latents, m)
    T, ST = O.project_orthogonal(P["U"], P["S"], P["s2"])
    Ty = T @ O.reshape_y(P["y"], n)
    np.linalg.cholesky(np.eye(256) * 2.0)      # BLAS thread-pool start-up is not part of the sample
    t0 = time.perf_counter()
    t_gram = 0.0
    for l in range(k):
        tg = time.perf_counter()
        mean, K = O.gp_mean_cov(P["gps"][l], P["x"])              # the same two steps as O.gp_logpdf, timed separately
        K[np.diag_indices_from(K)] += ST[l]
        t_gram += time.perf_counter() - tg
        O.gaussian_logpdf(mean, K, Ty[l])
    dt = time.perf_counter() - t0
    est = dt / k * m
    return {"value": 1.0 / est, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"oracle per-latent logpdf (NumPy Gram + LAPACK potrf/trtrs) for {k} of {m} latents at n={n}: "
                      f"{dt:.2f} s ({t_gram:.2f} s of it Gram assembly), extrapolated x{m}/{k}"}


def cpu_baseline_predictive(P):
    """Oracle posterior + marginals (LAPACK potrf, trtrs on the n x n* cross-Gram) for ONE latent at full size, extrapolated to
    the m independent latents."""
    from oracle import lmm_oracle as O
    m, n = P["m"], P["n"]
    T, ST = O.project_orthogonal(P["U"], P["S"], P["s2"])
    Ty0 = T[0] @ O.reshape_y(P["y"], n)
    xs = P["x"] + 0.5 * 20.0 / 575.0
    np.linalg.cholesky(np.eye(256) * 2.0)
    t0 = time.perf_counter()
    O.gp_mean_var(O.gp_posterior(P["gps"][0], P["x"], ST[0], Ty0), xs)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * m), "unit": "evals/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": f"oracle posterior + marginals of 1 of {m} latents at n = n* = {n}: {dt:.2f} s, extrapolated x{m}"}


def cpu_baseline_sampling(P):
    """Oracle prior sample of ONE latent at full n (NumPy Gram + LAPACK potrf + L z), extrapolated to the m independent latents."""
    from oracle import lmm_oracle as O
    m, n = P["m"], P["n"]
    z = np.random.default_rng(0).standard_normal(n)
    np.linalg.cholesky(np.eye(256) * 2.0)
    t0 = time.perf_counter()
    O.gp_rand(P["gps"][0], P["x"], 1e-4, z)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * m), "unit": "samples/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": f"oracle gp_rand (Gram + LAPACK potrf + L z, Float64) of 1 of {m} latents at n = {n}: {dt:.2f} s, extrapolated x{m}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="compute dtype of the per-latent matrices (lmm_set_compute_dtype); f64 is the parity mode")
    ap.add_argument("--proj", default="native", choices=["native", "bf16", "bf16x2"],
                    help="dtype of the H unprojection of predictive marginals (lmm_set_projection_dtype); bf16 = BASELINE configs[3] as named")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) through
        # torch.distributed.run.  This parent has not imported torch and never touches a GPU; it only relays the exit code.
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"[bench] --gpus {args.gpus} does not match WORLD_SIZE {world}: refusing to report a number for the wrong rank count")
    # Rehearsal switches (not used by the driver): LMM_BENCH_BACKEND=gloo reduces over CPU tensors and
    # LMM_BENCH_SHARE_GPU=1 puts every rank on GPU 0, so the N > 1 path can be exercised on a one-GPU box.
    backend = os.environ.get("LMM_BENCH_BACKEND", "nccl")
    if os.environ.get("LMM_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"[bench] rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                 "(one process per GPU; LMM_BENCH_SHARE_GPU=1 + LMM_BENCH_BACKEND=gloo rehearses N ranks on one GPU)")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        # no device_id: torch creates its NCCL communicator lazily at its first collective, i.e. AFTER the C ABI's communicator
        # below has been created and probed -- the two RCCL initialisations never interleave
        dist.init_process_group(backend)
    import lmm_amd
    from lmm_amd import _lib as L
    from lmm_amd.workloads import synthetic_problem      # input generation only; the oracle is imported by the cpu_baseline leg alone
    lmm_amd.init(local_rank)
    lmm_amd.set_compute_dtype(args.dtype)
    lmm_amd.set_projection_dtype(args.proj)
    dev = torch.device("cuda", local_rank)
    # The data-path collective is the C ABI's own RCCL communicator (lmm_allreduce_sum_f64: what a Julia / C caller binds);
    # torch.distributed is kept for the rendezvous (its store ships the RCCL unique id), the fence barrier and the max-over-ranks
    # clock.  select_collective creates and PROBES that communicator; if that fails on any rank, every rank falls back to
    # torch.distributed's all-reduce and the JSON line says why -- an N > 1 run reports a number either way.
    abi_comm, collective = lmm_amd.select_collective(backend, world)
    if world > 1 and rank == 0:
        print(f"[bench] collective: {collective}", file=sys.stderr, flush=True)
    from lmm_amd import model as lmm_model
    lmm_model.ILMM_ALLOW_DECOUPLED = (args.workload != "c1dense")

    m, p, n, kind, orth, desc = WORKLOADS[args.workload]
    s2 = 0.1
    P = synthetic_problem(m, p, n, kind, orth, s2=s2, seed=0)
    if args.workload == "notebook":      # x = 552 of 576 grid points on [0, 20]; S = singular values of rand(600, 20); sigma2 = 1e-6
        s2 = 1e-6
        keep = np.sort(np.random.default_rng(1).permutation(576)[:552])
        P["x"] = np.linspace(0.0, 20.0, 576)[keep]
        Usv, Ssv, _ = np.linalg.svd(np.random.default_rng(2).uniform(size=(p, m)), full_matrices=False)
        P["U"], P["S"], P["s2"] = np.ascontiguousarray(Usv), Ssv, s2
    fs, H = build_model(lmm_amd, P)
    xd = torch.from_numpy(P["x"]).to(dev)
    yd = torch.from_numpy(P["y"]).to(dev)
    xin = lmm_amd.MOInputIsotopicByOutputs(xd, p)
    if orth:
        shard = lmm_amd.latent_shard(m, rank, world)
        f = lmm_amd.ILMM(fs, H, shard=shard)
    else:
        shard = (0, m)                       # dense ILMM does not shard: replicas only
        f = lmm_amd.ILMM(fs, H)
    fx = f(xin, s2)
    rdev = dev if backend == "nccl" else torch.device("cpu")      # where the collectives' tensors live
    red = torch.zeros(1, dtype=torch.float64, device=rdev)
    red_host = np.zeros(1)
    stats = {"allreduce_s": 0.0, "nonfinite_partials": 0, "last_partial": None}      # this rank's diagnostics (per_rank in the JSON line)

    predictive = (args.workload == "c3")
    sampling = (args.workload == "c4")
    if sampling:
        normals = lmm_amd.DeviceNormals(1234 + 0)            # every rank draws the SAME normals (same seed, same stream order)
        jit_rand = (1e-9, 1e-4, 1e-4)

    def step_sampling():
        part = lmm_amd.rand(normals, fx, jitters=jit_rand, add_noise=(rank == 0))     # this rank's latents mixed through its H columns
        if world > 1:
            ta = time.perf_counter()
            if abi_comm:
                L.allreduce_sum(part)                        # ONE all-reduce of the n x p partial sample
            elif backend == "nccl":
                dist.all_reduce(part)
            else:
                pc = part.cpu(); dist.all_reduce(pc); part = pc
            stats["allreduce_s"] += time.perf_counter() - ta
        return float(part[0])

    if predictive:
        xs_in = lmm_amd.MOInputIsotopicByOutputs(xd + 0.5 * 20.0 / 575.0, p)      # test points between the training points

    def step_predictive():
        post = lmm_amd.posterior(fx, yd)                                            # this rank's latents only
        mean, var = lmm_amd.sharded_mean_and_var(post(xs_in, s2))                   # one all-reduce of 2 x p x n* doubles
        return float(mean[0])

    def step():
        if predictive:
            return step_predictive()
        if sampling:
            return step_sampling()
        part = lmm_amd.logpdf(fx, yd, rank == 0)
        stats["last_partial"] = part
        if not np.isfinite(part):
            stats["nonfinite_partials"] += 1
        if world > 1 and orth:
            ta = time.perf_counter()
            if abi_comm:
                red_host[0] = part
                L.allreduce_sum(red_host)    # ONE scalar RCCL all-reduce per evaluation, inside liblmm_hip.so
                out = float(red_host[0])
            else:
                red[0] = part
                dist.all_reduce(red)         # torch.distributed: gloo rehearsal, or nccl after an ABI-communicator failure
                out = float(red[0])
            stats["allreduce_s"] += time.perf_counter() - ta
            return out
        return part

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    val = None
    for _ in range(args.warmup):
        val = step()
    fence()
    stats["allreduce_s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val = step()
    dt_local = time.perf_counter() - t0          # this rank's own clock up to its last result (before the closing barrier)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt[0])
    steps = max(args.steps, 1)
    # --steps 0: roofline leg only (used for the rocprofv3 summary of the serial-stream pass); value is then null
    evals_per_s = (steps * (world if not orth else 1)) / dt if args.steps > 0 else None   # replicas: every rank runs the job

    roof = None
    extra = {}
    peak_tf = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else FP64_MFMA_PEAK_TFLOPS
    prof = None
    nprof = min(16, shard[1] - shard[0]) if orth else m     # one production-sized batch of latents (LMM_BATCH default 16)
    if (rank == 0 or world > 1) and not args.no_roofline and nprof > 0:
        # Instrumented pass: one more evaluation of one batch of this rank's latents with every launch of the hot kernels
        # bracketed by HIP events on its stream; the batch runs on ONE stream so that an event pair times its kernel alone (the
        # timed region above runs several batches on concurrent streams).  Rank 0's pass is the roofline leg (DESIGN.md
        # "Measurement"); at N > 1 every rank runs it on its own shard so that per_rank carries each rank's kernel class times.
        lib = lmm_amd.load()
        fprof = lmm_amd.ILMM(fs, H, shard=(shard[0], shard[0] + nprof))(xin, s2) if orth else fx
        L.check(lib.lmm_profile_begin(1))
        if sampling:
            lmm_amd.rand(lmm_amd.DeviceNormals(99), fprof, jitters=jit_rand, add_noise=False)
        elif predictive:
            lmm_amd.mean_and_var(lmm_amd.posterior(fprof, yd)(xs_in, s2))      # posterior + marginals of one batch, one stream
        else:
            lmm_amd.logpdf(fprof, yd, False)
        ent = (L.ProfEntryT * len(L.PROF_CLASSES))()
        L.check(lib.lmm_profile_end(ent))
        prof = {c: {"launches": int(ent[i].launches), "ms": float(ent[i].ms), "work": float(ent[i].work),
                    "bytes": float(ent[i].bytes)} for i, c in enumerate(L.PROF_CLASSES)}
    if rank == 0 and prof is not None:
        lib = lmm_amd.load()
        up = prof["update"]
        sv = prof.get("solve")
        if predictive and sv and sv["launches"] and sv["ms"] > 0:
            # configs[3]: per latent n^3/3 flops of factorisation against n* n^2 of the cross-solve R = K(x*, x) L^-T -- the block
            # updates of that triangular solve (trsm_rec) are the dominant kernel class of the posterior-predictive step
            ach = sv["work"] / (sv["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "gemm16p_kernel<DEPTH> (v_mfma_f64_16x16x4_f64, VGPR accumulators, software-pipelined) as the block updates "
                                               "R[:, c1] -= R[:, c0] L[c1, c0]' of the triangular solve K(x*, x) L^-T (trsm_rec: K = 64 .. n/2; + gemm16h_kernel on ragged rows)",
                    "achieved": round(ach, 3), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4), "traffic": None,
                    "launches": sv["launches"], "avg_launch_ms": round(sv["ms"] / sv["launches"], 4),
                    "flops_per_launch": sv["work"] / sv["launches"], "algorithmic_bytes_per_launch": sv["bytes"] / sv["launches"],
                    "mode": f"serial-stream instrumented pass over {nprof} latent(s): posterior + marginals"}
            if up["launches"] and up["ms"] > 0:
                extra["roofline_factor_update"] = {"bound": "mfma", "kernel": "potrf_node_kernel<2> (the K >= 1024 trailing updates of the n x n factorisations)",
                                                   "achieved": round(up["work"] / (up["ms"] * 1e-3) / 1e12, 3), "peak": peak_tf, "unit": "TFLOP/s",
                                                   "frac": round(up["work"] / (up["ms"] * 1e-3) / 1e12 / peak_tf, 4), "launches": up["launches"]}
            stp = prof.get("strip")
            if stp and stp["launches"] and stp["ms"] > 0:
                gbs = stp["bytes"] / (stp["ms"] * 1e-3) / 1e9
                extra["roofline_strip"] = {"bound": "hbm", "kernel": "strip_reduce_kernel (posterior mean and variance from one read of R)",
                                           "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                           "launches": stp["launches"]}
        elif up["launches"] and up["ms"] > 0:
            ach = up["work"] / (up["ms"] * 1e-3) / 1e12
            traffic = None       # HBM-side bytes per launch from the committed rocprofv3 --pmc passes (separate runs)
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload:
                    traffic = tj.get("update_kernel", {}).get("bytes_per_launch")
            roof = {"bound": "mfma", "kernel": ("gemm32w_kernel (v_mfma_f32_32x32x2_f32 SYRK/GEMM trailing update on 256 x 256 tiles, AccVGPR accumulators; gemm32_kernel<128> where those do not fill the device)" if args.dtype == "f32"
                                                else "potrf_node_kernel<2> (v_mfma_f64_16x16x4_f64, VGPR accumulators, software-pipelined SYRK/GEMM trailing update, K >= 1024; "
                                                     "one workgroup per matrix also factors the next panel's 128 x 128 diagonal block; the ragged last 64 rows are "
                                                     "half-height work items of the launch -- or gemm16h_kernel<true> behind it while several batches are in flight)"),
                    "achieved": round(ach, 3), "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": round(ach / peak_tf, 4), "traffic": traffic if args.dtype == "f64" else None,
                    "traffic_source": ("profiles/pmc_traffic.json: fabric bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                       "passes of this command (FETCH_SIZE x2, gfx950); NOT measured in this run") if traffic else None,
                    "launches": up["launches"], "avg_launch_ms": round(up["ms"] / up["launches"], 4),
                    "flops_per_launch": up["work"] / up["launches"],
                    "algorithmic_bytes_per_launch": up["bytes"] / up["launches"],
                    "mode": f"serial-stream instrumented pass over {nprof} latent(s)"}
        rg = prof.get("region")
        if roof is None and rg and rg["launches"] and rg["ms"] > 0:
            # small matrices (NC <= 1024): the whole factorisation is ONE potrf_region_kernel launch per batch -- the dominant kernel there
            ach = rg["work"] / (rg["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "potrf_region_kernel (whole blocked Cholesky of a batch in one dataflow launch: walker + helpers + row streams; latency-bound at these sizes)",
                    "achieved": round(ach, 4), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 5), "traffic": None,
                    "launches": rg["launches"], "avg_launch_ms": round(rg["ms"] / rg["launches"], 4), "flops_per_launch": rg["work"] / rg["launches"],
                    "mode": f"serial-stream instrumented pass over {nprof} latent(s)"}
        if roof is not None and rg and rg["launches"] and rg["ms"] > 0 and roof.get("kernel", "").startswith("potrf_node_kernel"):
            # the levels below K = 1024 (or 512) as dataflow launches: the base case of the recursion
            extra["roofline_region"] = {"bound": "mfma", "kernel": "potrf_region_kernel (a 512- or 1024-column block column per launch: walker + helpers + row streams)",
                                        "achieved": round(rg["work"] / (rg["ms"] * 1e-3) / 1e12, 3), "peak": peak_tf, "unit": "TFLOP/s",
                                        "frac": round(rg["work"] / (rg["ms"] * 1e-3) / 1e12 / peak_tf, 4), "launches": rg["launches"],
                                        "avg_launch_ms": round(rg["ms"] / rg["launches"], 4)}
        us = prof.get("update_short")
        if us and us["launches"] and us["ms"] > 0:
            extra["roofline_update_short"] = {"bound": "mfma", "kernel": "potrf_node_kernel<1> (the same fused update + leaf at K < 1024: the K = 512 updates between two dataflow launches, or every level of the panel recursion)",
                                               "achieved": round(us["work"] / (us["ms"] * 1e-3) / 1e12, 3), "peak": peak_tf, "unit": "TFLOP/s",
                                               "frac": round(us["work"] / (us["ms"] * 1e-3) / 1e12 / peak_tf, 4), "launches": us["launches"],
                                               "avg_launch_ms": round(us["ms"] / us["launches"], 4)}
        gr = prof["gram"]
        if gr["launches"] and gr["ms"] > 0:
            gbs = gr["work"] / (gr["ms"] * 1e-3) / 1e9
            # the kernel only WRITES (x is a few KB): the device's write-only rate, measured here by hipMemsetAsync into a block of the
            # same size as the batch's algorithmic bytes (capped at 8 GiB), is the ceiling this kernel can reach; `frac` stays against the
            # 8 TB/s spec (north_star's denominator), `frac_of_achievable` against the yardstick
            wr = C.c_double()
            wbytes = int(min(max(gr["work"], 1 << 26), 8 << 30))
            have_wr = lib.lmm_dev_write_rate(C.c_size_t(wbytes), 3, C.byref(wr)) == 0 and wr.value > 0
            extra["roofline_gram"] = {"bound": "hbm", "kernel": "gram_kernel (lower-triangular f64 write)",
                                      "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                      "achievable_write_gbs": round(wr.value, 1) if have_wr else None,
                                      "frac_of_achievable": round(gbs / wr.value, 4) if have_wr else None,
                                      "achievable_source": f"hipMemsetAsync of {wbytes >> 20} MiB x3 in this process (lmm_dev_write_rate)",
                                      "matrices": gr["launches"], "avg_ms_per_matrix": round(gr["ms"] / gr["launches"], 4),
                                      "note": "the matrices of a batch are assembled by one launch (blockIdx.z = latent)"}
        extra["kernel_classes_ms"] = {c: round(v["ms"], 3) for c, v in prof.items()}
        tf = C.c_double()
        if lib.lmm_dev_mfma_f64_peak(C.byref(tf)) == 0:
            extra["mfma_f64_16x16x4_issue_rate_measured_tflops"] = round(tf.value, 2)
        fl = m * n ** 3 / 3.0 if orth else (m * n) ** 3 / 3.0
        if evals_per_s:
            extra["end_to_end_cholesky_tflops"] = round(fl * evals_per_s / 1e12 / (1 if orth else world), 3)
        if args.workload == "c2" and world == 1 and orth and args.steps > 0:
            # what ONE rank of the 8-GPU job does per evaluation: its 4-latent share (with the regulariser, as rank 0), timed here
            # on one GPU -- the per-rank time the driver's N = 8 run should show, before the 8-byte all-reduce
            fsh = lmm_amd.ILMM(fs, H, shard=lmm_amd.latent_shard(m, 0, 8))(xin, s2)
            lmm_amd.logpdf(fsh, yd, True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                lmm_amd.logpdf(fsh, yd, True)
            torch.cuda.synchronize(); dts = (time.perf_counter() - t1) / 3
            tfs = (m // 8) * n ** 3 / 3.0 / dts / 1e12
            extra["share_of_8gpu_job"] = {"latents": m // 8, "ms_per_eval": round(dts * 1e3, 2), "cholesky_tflops": round(tfs, 2),
                                          "frac_of_fp64_peak": round(tfs / FP64_MFMA_PEAK_TFLOPS, 4),
                                          "implied_speedup_at_8_gpus": round((dt / steps) / dts, 2)}

    # per-rank diagnostics: when an N > 1 run misses its scaling target the line must say which rank or phase was slow
    mine = {"rank": rank, "latents": [int(shard[0]), int(shard[1])], "ms_per_step": dt_local / steps * 1e3,
            "allreduce_ms_per_step": stats["allreduce_s"] / steps * 1e3, "last_partial": stats["last_partial"],
            "nonfinite_partials": stats["nonfinite_partials"],
            "kernel_classes_ms": ({c: round(v["ms"], 3) for c, v in prof.items()} if prof else None),
            "instrumented_latents": nprof if prof else 0}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    bad = sum(r["nonfinite_partials"] for r in per_rank)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_predictive(P) if predictive else (cpu_baseline_sampling(P) if sampling else cpu_baseline(P, orth))

    if rank == 0:
        line = {
            "metric": ("posterior + marginals evals/sec" if predictive else ("prior samples/sec" if sampling else "logpdf evals/sec")) +
                      ("" if orth or world == 1 else " (independent replicas: the dense-H path does not shard)"),
            "value": evals_per_s, "unit": "evals/s",
            "obs_per_s": evals_per_s * n * p if evals_per_s else None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if orth else "replicas",
            "vs_baseline": (evals_per_s / NOTEBOOK_PUBLISHED_EVALS_PER_S) if (args.workload == "notebook" and evals_per_s) else None,
            "dtype": args.dtype + ("" if args.proj == "native" else f"+{args.proj}proj"), "data": "synthetic",
            "config": {"workload": desc, "m": m, "p": p, "n": n, "kernel": kind, "sigma2": s2,
                       "latents_per_gpu": (shard[1] - shard[0]), "parallelism": f"latent-shard x{world}" if orth else "replicas",
                       "collective": collective, "rccl_ranks": L.comm_world(),
                       "projection_dtype": args.proj},
            ("first_predictive_mean" if predictive else ("first_sample_value" if sampling else "logpdf")): val,
            "roofline": roof, "cpu_baseline": cpu, "per_rank": per_rank,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        if abi_comm:
            L.comm_destroy()
        dist.destroy_process_group()
    if bad:
        sys.exit(f"[bench] rank {rank}: {bad} non-finite logpdf partial(s) over the ranks: the value above is not a measurement")


if __name__ == "__main__":
    main()
