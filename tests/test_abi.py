"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/lmm_hip.h declares,
the host-only entry point works without a GPU, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import lmm_amd
from lmm_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lmm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lmm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = lmm_amd.load()
    decl = declared_symbols()
    assert len(decl) >= 20
    for s in decl:
        assert hasattr(lib, s), f"liblmm_hip.so does not export {s}"
    assert sorted(L.SYMBOLS) == decl, "SYMBOLS list in _lib.py is out of sync with include/lmm_hip.h"


def test_struct_layouts_match_header():
    assert C.sizeof(L.GpT) == 32 and L.GpT.variance.offset == 8 and L.GpT.mean.offset == 24
    assert C.sizeof(L.JittersT) == 24
    assert C.sizeof(L.ProfEntryT) == 32 and C.sizeof(L.GpGradT) == 24


def test_orthogonal_validate_host_only():
    """reference test/orthogonal_matrix.jl:1-14."""
    rng = np.random.default_rng(0)
    with pytest.raises(ValueError, match="`U` is not an orthogonal matrix"):
        lmm_amd.Orthogonal(rng.uniform(size=(5, 3)), rng.uniform(size=3))
    U, S, _ = np.linalg.svd(rng.uniform(size=(5, 3)), full_matrices=False)
    H = lmm_amd.Orthogonal(U, S)
    assert H.shape == U.shape
    np.testing.assert_array_equal(H.S, S)
    np.testing.assert_array_equal(H.U, U)
    np.testing.assert_allclose(H.collect(), U @ np.diag(np.sqrt(S)))
    lmm_amd.Orthogonal(rng.uniform(size=(5, 3)), S, validate_fields=False)     # validate_fields kwarg


def test_helpers_match_reference_utils():
    """reference test/ilmm.jl:55-71."""
    assert lmm_amd.noise_var(2) == 2
    y = np.random.default_rng(1).uniform(size=16)
    assert lmm_amd.reshape_y(y, 8).shape == (2, 8)
    assert lmm_amd.reshape_y(y, 2).shape == (8, 2)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern32Kernel())])
    H = np.random.default_rng(2).uniform(size=(2, 1))
    x = lmm_amd.MOInputIsotopicByOutputs(np.random.default_rng(3).uniform(size=(2, 2)), 2)   # ColVecs(rand(2,2))
    ilmm = lmm_amd.ILMM(fs, H)
    f, H2, s2, xx = lmm_amd.unpack(ilmm(x, 0.1))
    assert f is fs and H2 is ilmm.H and s2 == 0.1 and xx is x.x
    assert lmm_amd.get_latent_gp(ilmm) == fs
    assert lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.SEKernel())]) == lmm_amd.IndependentMOGP([lmm_amd.GP(lmm_amd.SEKernel())])
    with pytest.raises(RuntimeError, match="out dim of x != out dim of f"):
        lmm_amd.unpack(ilmm(lmm_amd.MOInputIsotopicByOutputs(np.zeros(3), 3), 0.1))


def test_no_cpu_fallback():
    """Without a GPU every compute entry point must raise; nothing routes through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.SEKernel())])
    fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(np.array([[1.0], [0.0]]), np.array([1.0])))(
        lmm_amd.MOInputIsotopicByOutputs(np.arange(4.0), 2), 0.1)
    with pytest.raises(lmm_amd.LMMError, match="no HIP device"):
        lmm_amd.logpdf(fx, np.zeros(8))
    src = "".join(open(os.path.join(ROOT, "linearmixingmodels.jl_amd", f)).read()
                  for f in ("__init__.py", "_lib.py", "model.py", "parallel.py"))
    assert "import oracle" not in src and "from oracle" not in src and "lmm_oracle" not in src


def test_latent_shard_partition():
    for m in (1, 3, 8, 32, 33):
        for world in (1, 2, 3, 4, 8):
            blocks = [lmm_amd.latent_shard(m, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_reorder_indices_known_answers():
    """reference test/independent_mogp.jl:86-98."""
    x = lmm_amd.MOInputIsotopicByFeatures(np.linspace(0.0, 2.0, 3), 2)
    v_by_output, v_by_features = np.array([1, 1, 1, 2, 2, 2]), np.array([1, 2, 1, 2, 1, 2])
    o2f = lmm_amd.indices_which_reorder_outputs_to_features(x) - 1
    f2o = lmm_amd.indices_which_reorder_features_to_outputs(x) - 1
    np.testing.assert_array_equal(v_by_output[o2f], v_by_features)
    np.testing.assert_array_equal(v_by_features[f2o], v_by_output)
    np.testing.assert_array_equal(v_by_output[o2f][f2o], v_by_output)
    np.testing.assert_array_equal(v_by_features[f2o][o2f], v_by_features)


def test_bench_refuses_a_rank_count_it_is_not_running():
    """bench.py --gpus N must not report a number for another rank count: under a launcher that set WORLD_SIZE != N it exits
    non-zero before touching a GPU (with WORLD_SIZE unset it starts its own N ranks through torch.distributed.run)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "does not match WORLD_SIZE" in out.stderr
    assert out.stdout.strip() == ""


def test_select_collective_branches():
    """parallel.select_collective (what bench.py --gpus N and the sharded_* helpers use to pick the N > 1 all-reduce):
    backend nccl -> the C ABI's RCCL communicator is created AND probed, and is the data-path collective; a failure of either step
    on any rank -> every rank drops it, torch.distributed carries the reduce and the description says why; gloo -> torch.
    (RCCL itself needs one GPU per rank; the first hardware N > 1 run is the driver's.)"""
    import lmm_amd
    from lmm_amd import parallel as PP
    calls = []

    def ok_init():
        calls.append("init"); return 2

    def ok_probe(world):
        calls.append(("probe", world))

    def agree(ok):
        calls.append(("agree", ok)); return ok

    def destroy():
        calls.append("destroy")

    assert PP.select_collective("nccl", 1) == (False, None)
    assert PP.select_collective("gloo", 2) == (False, "torch.distributed/gloo")
    assert PP.select_collective("nccl", 2, ok_init, ok_probe, agree, destroy) == (True, PP.ABI_COLLECTIVE)
    assert calls == ["init", ("agree", True), ("probe", 2), ("agree", True)]      # the init outcome is agreed BEFORE the probe collective
    # the communicator comes up with the wrong size
    calls.clear()
    use, desc = PP.select_collective("nccl", 4, ok_init, ok_probe, agree, destroy)
    assert not use and "ABI RCCL failed" in desc and "2 ranks, expected 4" in desc and calls[-1] == "destroy"
    # init raises (e.g. LMM_ERR_RCCL from ncclCommInitRank)

    def bad_init():
        raise lmm_amd.LMMError("ncclCommInitRank failed: unhandled system error")

    calls.clear()
    use, desc = PP.select_collective("nccl", 2, bad_init, ok_probe, agree, destroy)
    assert not use and desc.startswith("torch.distributed/nccl (ABI RCCL failed: LMMError: ncclCommInitRank failed")
    assert ("probe", 2) not in calls          # a rank whose init failed never enters the probe all-reduce ...
    # ... and neither does a rank whose own init succeeded when ANOTHER rank's did not (it would block in RCCL for ever)
    calls.clear()
    use, desc = PP.select_collective("nccl", 2, ok_init, ok_probe, lambda ok: (calls.append(("agree", ok)), False)[1], destroy)
    assert not use and ("probe", 2) not in calls and "on another rank" in desc
    # the first collective fails

    def bad_probe(world):
        raise RuntimeError("ABI all-reduce probe returned 1.0")

    use, desc = PP.select_collective("nccl", 2, ok_init, bad_probe, agree, destroy)
    assert not use and "probe returned" in desc
    # this rank is fine but another one is not: still fall back (all ranks must take the same branch)
    use, desc = PP.select_collective("nccl", 2, ok_init, ok_probe, lambda ok: False, destroy)
    assert not use and "on another rank" in desc


def test_bench_uses_select_collective():
    """bench.py must route its N > 1 collective choice through select_collective (no unguarded comm_init_from_torch)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    assert "select_collective(backend, world)" in src and "comm_init_from_torch" not in src
    assert '"per_rank": per_rank' in src and "rccl_ranks" in src


def test_check_info_translation_and_ordering():
    """lmm_dev_check_info (host only): the pivot-info words of a batch -> status.  A dependency-wait timeout marker of
    potrf_region_kernel (-7777) in ANY position outranks a PosDefException of an earlier latent and becomes LMM_ERR_HIP; a
    non-zero pivot alone is LMM_ERR_NOT_PD with latent and pivot in lmm_last_error_detail; all zeros is LMM_OK."""
    import ctypes as C
    import lmm_amd
    from lmm_amd import _lib as L
    lib = lmm_amd.load()

    def status(words, begin=0):
        arr = (C.c_int * len(words))(*words)
        return lib.lmm_dev_check_info(arr, len(words), begin)

    assert status([0, 0, 0]) == L.LMM_OK
    assert status([0, 17, 0, 3], begin=4) == L.LMM_ERR_NOT_PD
    lat, info = C.c_int(), C.c_int()
    lib.lmm_last_error_detail(C.byref(lat), C.byref(info))
    assert (lat.value, info.value) == (5, 17)
    assert status([-7777]) == L.LMM_ERR_HIP
    assert b"timed out" in lib.lmm_last_error_string()
    assert status([9, 0, -7777, 0]) == L.LMM_ERR_HIP            # the timeout is not masked by the earlier non-PD pivot
    assert b"latent 2" in lib.lmm_last_error_string()
    with pytest.raises(lmm_amd.LMMError):
        L.check(status([0, -7777]))
    with pytest.raises(lmm_amd.PosDefException):
        L.check(status([0, 2]))


def test_region_plan_covers_the_rows_and_is_deterministic():
    """lmm_dev_region_plan (host arithmetic of launch_region, no GPU): whatever split of 128- / 64-row tiles the list-scheduling
    estimate picks, the row tasks cover every row below the square exactly once, the assistants are all or none, and the same
    shape gets the same plan again (the plan is cached per shape and feeds the kernel's task decode)."""
    import ctypes as C
    from lmm_amd import _lib as L
    lib = L.load()
    lib.lmm_dev_region_plan.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_int)]
    seen = set()
    for P in (1, 2, 5, 8):
        for nb in (1, 3, 4, 8, 16, 32):
            for rows_below, real in [(0, 0), (64, 1), (64, 64), (192, 130), (320, 257), (1088, 1025), (3136, 3073), (7232, 7169), (15424, 15361)]:
                na = max(0, 2 * P - 6)
                out = (C.c_int * 3)()
                assert lib.lmm_dev_region_plan(P, nb, rows_below, real, 256, na, out) == L.LMM_OK
                n128, tasks, used = out[0], out[1], out[2]
                t128 = (rows_below + 127) // 128
                assert 0 <= n128 <= t128 and tasks >= n128 and used in (0, na)
                rest = rows_below - 128 * n128
                if n128 == t128:
                    assert tasks == t128                                  # all tall: the last tile may be a half tile
                else:
                    assert rest > 0 and tasks - n128 == (rest + 63) // 64 and rows_below % 64 == 0
                again = (C.c_int * 3)()
                assert lib.lmm_dev_region_plan(P, nb, rows_below, real, 256, na, again) == L.LMM_OK
                assert list(again) == list(out)
                seen.add((n128 == t128, n128 == 0))
    assert (True, False) in seen and (False, True) in seen               # both pure forms occur; mixed ones are shape dependent
    # the plan cache is keyed on the CU count too: the same shape on a smaller device must be planned for THAT device --
    # the split of 128- / 64-row tiles that balances 256 CUs is not the one that balances 120; the second call is planned afresh, not replayed
    big, small, big2 = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
    assert lib.lmm_dev_region_plan(8, 4, 15424, 15361, 256, 10, big) == L.LMM_OK
    assert lib.lmm_dev_region_plan(8, 4, 15424, 15361, 120, 10, small) == L.LMM_OK
    assert lib.lmm_dev_region_plan(8, 4, 15424, 15361, 256, 10, big2) == L.LMM_OK
    assert list(big) == list(big2) and list(small) != list(big)           # (a cache keyed without the CU count replays the first plan)
    out = (C.c_int * 3)()
    assert lib.lmm_dev_region_plan(9, 4, 128, 128, 256, 0, out) == L.LMM_ERR_ARG
