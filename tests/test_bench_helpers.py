"""CPU suite: the pieces of bench.py and distributed.py that need no GPU -- counter files tied to a build, the
interior / boundary split of a rank's blocks."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_counter_files_of_another_build_are_refused(tmp_path):
    import bench
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps({"kernel": "panel_kernel<double", "alg_bytes_per_launch": 100, "build": "abc",
                             "traffic_bytes_per_launch": 105}))
    d, why = bench.counter_file(str(f), "abc", kernel="bsm::panel_kernel<double,8,true,false,true>", alg_bytes=100)
    assert d is not None and why is None and d["traffic_bytes_per_launch"] == 105
    d, why = bench.counter_file(str(f), "xyz", kernel="bsm::panel_kernel<double,8,true,false,true>", alg_bytes=100)
    assert d is None and "build" in why
    d, why = bench.counter_file(str(f), "abc", kernel="bsm::panel_kernel<float,8,true,false,true>", alg_bytes=100)
    assert d is None and "kernel" in why
    d, why = bench.counter_file(str(f), "abc", kernel="bsm::panel_kernel<double,8,true,false,true>", alg_bytes=101)
    assert d is None and "bytes" in why
    d, why = bench.counter_file(str(tmp_path / "missing.json"), "abc")
    assert d is None and "no counter file" in why
    # the committed files belong to the committed kernels: same build id as the library reports
    from bsm_amd import _lib
    build = _lib.lib().bsm_version().decode().split("build ")[-1]
    for name in (bench.PMC_FILE, bench.MFMA_FILE):
        if os.path.exists(name):
            assert json.load(open(name))["build"] == build, name + " was taken with another build of the kernels: re-run tools/profile_round.sh"


def test_interior_boundary_split():
    import bsm_amd as bsm
    from bsm_amd import distributed as D
    # symmetric, banded: the boundary blocks of a rank are the off-diagonal blocks that reach into the previous rank
    prob = bsm.synthetic.config5(n=6000, lo=16, hi=96, halfband=3)
    seen_d = seen_o = 0
    for r in range(3):
        local, own, touched = D.split_symmetric(prob, r, 3)
        inner, outer, bt, bx = D.split_interior(local, own)
        seen_d += len(inner["diagonals"]) + len(outer["diagonals"])
        seen_o += len(inner["offdiagonals"]) + len(outer["offdiagonals"])
        for idx in inner["diagonalindices"] + inner["rowindices"] + inner["colindices"]:
            assert own[0] <= idx.min() and idx.max() <= own[1]
        for ri, ci in zip(outer["rowindices"], outer["colindices"]):
            assert not (own[0] <= min(ri.min(), ci.min()) and max(ri.max(), ci.max()) <= own[1])
        if r == 0:
            assert D.is_empty(outer) and bt[1] < bt[0] and bx[1] < bx[0]
        else:
            assert not D.is_empty(outer) and bt[0] < own[0] <= bt[1] and bx == bt  # symmetric: reads what it writes
            assert (bt[0], bt[1]) == (touched[0], max(idx.max() for idx in outer["rowindices"]))
    assert seen_d == len(prob["diagonals"]) and seen_o == len(prob["offdiagonals"])
    # VBCRS with scattered columns: interior = the blocks whose columns the rank owns
    vp = bsm.synthetic.config2(n=8000, nblocks=400)
    total = 0
    for r in range(4):
        local, own = D.split_vbcrs(vp, r, 4)
        inner, outer, bt, bx = D.split_interior(local, own)
        total += len(inner["blocks"]) + len(outer["blocks"])
        for c0, b in zip(inner["colstart"], inner["blocks"]):
            assert own[0] <= c0 and c0 + b.shape[1] - 1 <= own[1]
        for c0, b in zip(outer["colstart"], outer["blocks"]):
            assert c0 < own[0] or c0 + b.shape[1] - 1 > own[1]
        if len(outer["blocks"]):
            assert own[0] <= bt[0] and bt[1] <= own[1]  # they write own rows only, but read beyond
    assert total == len(vp["blocks"])


def test_live_traffic_falls_back_when_the_profiler_is_absent(monkeypatch):
    """bench.live_traffic: no rocprofv3 on the box -> (None, reason); the caller then reads the stamped file"""
    import shutil
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    d, why = bench.live_traffic("panel_kernel<double, 8, true, false", 54553920)
    assert d is None and "rocprofv3" in why


def test_live_traffic_is_skipped_under_a_profiler(monkeypatch):
    """bench.py started under rocprofv3 (its preload / control variables in the environment) must not start profiler
    children of its own: they would inherit the preload and nest profilers"""
    import shutil
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: "/opt/rocm/bin/rocprofv3")
    for k, v in (("ROCPROFILER_LIBRARY_CTOR", "1"), ("ROCP_TOOL_LIBRARIES", "x"), ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")):
        monkeypatch.setenv(k, v)
        d, why = bench.live_traffic("panel_kernel<double, 8, true, false", 54553920)
        assert d is None and "profiler" in why and k in why
        monkeypatch.delenv(k)
    assert bench.profiler_in_environment({"LD_PRELOAD": "/usr/lib/libjemalloc.so", "PATH": "/opt/rocm/bin"}) is None


def test_fixture_coo_of_the_bench_is_the_reference_operator():
    """bench.fixture_coo (the parity reference of the BEM legs) against the test suite's own COO product, per element type"""
    import numpy as np
    import bench
    from _common import fixture_problem, scipy_mul, relerr, N
    for tname, dtn, part, tol in bench.BEM_TYPES:
        fx = fixture_problem("cuboid", np.dtype(dtn), part)
        n0 = fx["size"][0]
        rng = np.random.default_rng(3)
        x = rng.standard_normal(n0).astype(np.complex128 if np.dtype(dtn).kind == "c" else np.float64)
        got = bench.fixture_coo(np, fx, n0) @ x
        ref = scipy_mul({k: ([b.astype(x.dtype) for b in v] if k in ("diagonals", "offdiagonals") else v) for k, v in fx.items()},
                        N, x, np.zeros(n0, x.dtype))
        assert relerr(got, ref) < 1e-13, tname
