"""CPU suite: the packed image the HIP kernels walk (interpreted by tests/_common.py) against the
oracle on seeded random operators (tests/_fuzz.py) -- the host analysis on layout corner cases."""
import numpy as np
import pytest

from _common import Cc, N, T, interpret_image, oracle_mul, rand_vec
from _fuzz import GEN, seed_of

NODEV = -2


@pytest.fixture(scope="module")
def env():
    import bsm_amd as bsm
    from oracle import load_oracle
    return bsm, load_oracle()


@pytest.mark.parametrize("kind", ["blocksparse", "vbcrs", "symmetric"])
@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_random_operators_interpreted_image_matches_the_oracle(env, kind, dtype):
    bsm, oracle = env
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(seed_of(kind, dtype) + 1)
    for case in range(8):
        p = GEN[kind](rng, dtype)
        modes = ["auto", "atomic", "gather"] + (["colored"] if kind != "vbcrs" else [])
        acc = modes[case % len(modes)]
        kw = {"accumulate": acc}
        timg = kind != "symmetric" and case % 3 == 0
        if timg:
            kw["transpose_image"] = True
        try:
            A = bsm.synthetic.build(p, device=NODEV, **kw)
        except RuntimeError as e:
            assert acc == "colored" and "repeat" in str(e), (kind, dtype, case, str(e))
            continue
        nr, nc = p["size"]
        for op in (N, T, Cc):
            if op == Cc and dtype.kind != "c":
                continue
            xl, yl = (nc, nr) if op == N else (nr, nc)
            x, y0 = rand_vec(rng, xl, dtype), rand_vec(rng, yl, dtype)
            ref = oracle_mul(oracle, p, op, x, y0, -0.5, 1.25, False)
            got = interpret_image(A, op, x, y0, -0.5, 1.25, False, timage=(timg and op != N))
            scale = max(np.max(np.abs(ref)), 1e-30)
            assert np.max(np.abs(got - ref)) / scale < 1e-12, (kind, dtype, case, acc, op)


@pytest.mark.parametrize("kind", ["blocksparse", "symmetric"])
def test_random_operators_coarser_wave_records_match_the_oracle(env, kind, monkeypatch):
    """The second wave-record list the multi-RHS kernels walk (bsm_get_image 8), forced onto small random operators
    (2 KB per wave against waves of 256 bytes): same products through the image interpreter."""
    from _common import get_image
    bsm, oracle = env
    monkeypatch.setenv("BSM_WAVE_BYTES", "256")
    monkeypatch.setenv("BSM_MULTI_WAVE_BYTES", "2048")
    dtype = np.dtype(np.float64)
    rng = np.random.default_rng(seed_of(kind, dtype) + 7)
    seen = 0
    for case in range(10):
        p = GEN[kind](rng, dtype)
        A = bsm.synthetic.build(p, device=NODEV, accumulate="atomic")
        if len(get_image(A, multi=True)[3]) == 0:
            continue
        seen += 1
        nr, nc = p["size"]
        for op in (N, T):
            xl, yl = (nc, nr) if op == N else (nr, nc)
            x, y0 = rand_vec(rng, xl, dtype), rand_vec(rng, yl, dtype)
            ref = oracle_mul(oracle, p, op, x, y0, -0.5, 1.25, False)
            got = interpret_image(A, op, x, y0, -0.5, 1.25, False, multi=True)
            scale = max(np.max(np.abs(ref)), 1e-30)
            assert np.max(np.abs(got - ref)) / scale < 1e-12, (kind, case, op)
    assert seen >= 3
