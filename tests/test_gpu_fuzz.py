"""GPU suite: seeded random operators (tests/_fuzz.py) of every type / element type / accumulation
mode, HIP path through the C ABI against the CPU oracle."""
import numpy as np
import pytest

from _common import Cc, N, T, oracle_mul, rand_vec
from _fuzz import GEN, seed_of

pytestmark = pytest.mark.gpu
TOL = {np.dtype(np.float64): 1e-12, np.dtype(np.complex128): 1e-12,
       np.dtype(np.float32): 1e-5, np.dtype(np.complex64): 1e-5}


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import bsm_amd as bsm
    from bsm_amd import _lib as L
    from oracle import load_oracle
    L.lib()
    return torch, bsm, load_oracle()


@pytest.mark.parametrize("kind", ["blocksparse", "vbcrs", "symmetric"])
@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32, np.complex64])
def test_random_operators_match_the_oracle(env, kind, dtype):
    torch, bsm, oracle = env
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(seed_of(kind, dtype))
    for case in range(30):
        p = GEN[kind](rng, dtype)
        modes = ["auto", "atomic", "gather"]
        if kind != "vbcrs":
            modes.append("colored")
        acc = modes[case % len(modes)]
        kw = {"accumulate": acc}
        if kind != "symmetric" and case % 3 == 0:
            kw["transpose_image"] = True
        try:
            A = bsm.synthetic.build(p, **kw)
        except RuntimeError as e:  # coloured mode refuses repeated indices inside a row set: documented
            assert acc == "colored" and "repeat" in str(e), (kind, dtype, case, str(e))
            continue
        nr, nc = p["size"]
        for op in (N, T, Cc):
            if op == Cc and dtype.kind != "c":
                continue
            xl, yl = (nc, nr) if op == N else (nr, nc)
            x, y0 = rand_vec(rng, xl, dtype), rand_vec(rng, yl, dtype)
            for alpha, beta, strong in ((1, 0, True), (-0.5, 1.25, False)):
                ref = oracle_mul(oracle, p, op, x, y0, alpha, beta, strong)
                Aop = A if op == N else (bsm.transpose(A) if op == T else bsm.adjoint(A))
                yd = torch.from_numpy(y0.copy()).cuda()
                bsm.mul(yd, Aop, torch.from_numpy(x).cuda(), alpha, False if strong else beta)
                got = yd.cpu().numpy()
                scale = max(np.max(np.abs(ref)), 1e-30)
                assert np.max(np.abs(got - ref)) / scale < TOL[dtype], (kind, dtype, case, acc, op, alpha, beta)
            if case % 2 == 1:  # A*X with a column-major matrix (bsm_mul_multi): passes of 8, padded remainders
                k = int(rng.integers(2, 18))
                X = np.asfortranarray(np.stack([rand_vec(rng, xl, dtype) for _ in range(k)], axis=1))
                Y0 = np.asfortranarray(np.stack([rand_vec(rng, yl, dtype) for _ in range(k)], axis=1))
                Yd = torch.from_numpy(Y0.T.copy()).cuda().T  # column-major device matrix
                # (complex scalars for the complex types: the ComplexF64 8-column pass folds alpha into its x rows)
                am, bm = (-0.5 + 0.75j, 1.25 - 0.5j) if dtype.kind == "c" else (-0.5, 1.25)
                bsm.mul(Yd, Aop, torch.from_numpy(X.T.copy()).cuda().T, am, bm)
                got = Yd.cpu().numpy()
                for j in range(k):
                    ref = oracle_mul(oracle, p, op, X[:, j].copy(), Y0[:, j].copy(), am, bm, False)
                    scale = max(np.max(np.abs(ref)), 1e-30)
                    assert np.max(np.abs(got[:, j] - ref)) / scale < TOL[dtype], (kind, dtype, case, acc, op, "multi", j)


@pytest.mark.parametrize("env_extra", [{"BSM_MULTI_IL": "2"}, {"BSM_MULTI_IL": "2", "BSM_IL_XCD": "5"},
                                       {"BSM_MULTI_IL": "2", "BSM_IL_XCD": "0"}, {"BSM_MULTI_IL": "0"}])
def test_multi_rhs_fuzz_with_the_interleaved_pass_forced_and_switched_off(env_extra):
    """The interleaved multi-RHS pass is chosen per image (csrc/bsm_kernels.hip: il_applies) from switches read once per
    process.  In child processes: the pass FORCED onto every image that accumulates with atomics (tall panels,
    transposed-only and forward-only products included; tall panels with all their row blocks in flight and the workgroups
    dealt to the XCDs in runs of 16), the same in runs of 5 workgroups (grids padded to multiples of 40) and in the plain
    order, and the pass switched OFF (the round-4 kernels) -- random operators of the three types and four element types, every
    accumulation mode, ops N / T / C, 2-35 columns, every column against the oracle (tools/fuzz_multi_seeds.py, one seed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_multi_seeds.py"), "4100", "1"], env=dict(os.environ, **env_extra),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "MISMATCH" not in out, out[-2000:] + r.stderr.decode()[-2000:]
    last = [ln for ln in out.splitlines() if ln.startswith("seed ")][-1]
    assert last.endswith(" 0 mismatches") and int(last.split(":")[1].split()[0]) > 1000, last
