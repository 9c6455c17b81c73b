"""The C ABI used from plain C (examples/c_api_demo.c): what any FFI (Julia ccall, cgo, JNI) sees.
CPU: builds with gcc, links libbsmrocm.so, checks the bookkeeping KATs on an analysis-only handle.
GPU: additionally runs the SURVEY 8c product KATs through bsm_mul with host buffers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from bsm_amd import _lib
    _lib.lib()  # make sure libbsmrocm.so exists (compiled on demand)
    exe = str(tmp_path / "c_api_demo")
    libdir = os.path.join(ROOT, "blocksparsematrices.jl_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_api_demo.c"), "-L", libdir, "-lbsmrocm",
                           f"-Wl,-rpath,{libdir}", "-o", exe])
    return exe


def test_c_demo_bookkeeping(tmp_path, bsm):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "perm       = 3 2 1 4" in out.stdout and "rowptr     = 1 3 5" in out.stdout
    assert "partition  = 1 0 0 1, own = [1,4] [5,6]" in out.stdout
    assert out.stdout.strip().endswith("OK")


@pytest.mark.gpu
def test_c_demo_products_on_gpu(tmp_path, bsm):
    out = subprocess.run([_build(tmp_path), "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "BlockSparseMatrix KAT y = [10 0 27 0]" in out.stdout
    assert "SymmetricBlockMatrix KAT y = [11 14 1 2]" in out.stdout
    assert "two-device SymmetricBlockMatrix KAT y = [11 14 1 2]" in out.stdout
    assert "rowcolvals: 5 triples" in out.stdout


@pytest.mark.gpu
def test_partitioned_vectors_from_compiled_code(tmp_path, bsm):
    """examples/parts_demo.cpp: bsm_mul_parts with raw hipMalloc'ed vector parts on a context of two (virtual) devices,
    the y parts of one product as the x parts of the next -- compiled with hipcc against the C ABI only."""
    from bsm_amd import _lib
    _lib.lib()
    exe = str(tmp_path / "parts_demo")
    libdir = os.path.join(ROOT, "blocksparsematrices.jl_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "parts_demo.cpp"), "-L", libdir, "-lbsmrocm",
                           f"-Wl,-rpath,{libdir}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "partitioned SymmetricBlockMatrix KAT y = [11 14 1 2], A*y = [5 98 11 22]" in out.stdout
    assert out.stdout.strip().endswith("OK")
