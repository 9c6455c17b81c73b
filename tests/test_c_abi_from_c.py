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


def test_struct_layouts_match_the_header_and_the_bindings(tmp_path, bsm):
    """include/bsm_rocm.h pins the layout of its three structs with static assertions; the same numbers are asserted
    here against the ctypes mirror (blocksparsematrices.jl_amd/_lib.py) and against a C program compiled from the
    header, and they are the ones the Julia mirror (julia/BlockSparseMatricesROCm.jl: BsmOptions, BsmPartInfo) is
    written to -- that file cannot be executed here, so a drift has to fail on this side."""
    import ctypes as C
    import re
    from bsm_amd import _lib as L
    expect = {"bsm_options": (72, dict(struct_size=0, device=4, scheduler=8, accumulate=12, validate=16, transpose_image=20,
                                       own_lo=24, own_hi=32, ctx=40, blocks_memspace=48, coloring=56, reserved=64)),
              "bsm_part_info_t": (88, dict(device=0, own_lo=8, own_hi=16, touched_lo=24, touched_hi=32, device_bytes=40,
                                           nblocks=48, col_lo=56, col_hi=64, reserved=72)),
              "bsm_stats_t": (128, dict(nnz=0, stored_entries=8, alg_bytes=16, device_bytes=24, npanels=32, ntasks=40,
                                        nworkgroups=48, exclusive=56, win_emissions=64, win_inside=72, win_flushed=80,
                                        reserved=88))}
    mirror = {"bsm_options": L.BsmOptions, "bsm_part_info_t": L.BsmPartInfo, "bsm_stats_t": L.BsmStats}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "bsm_rocm.h"', "int main(void) {"]
    for name, (size, offs) in expect.items():
        assert C.sizeof(mirror[name]) == size, name
        for f, o in offs.items():
            assert getattr(mirror[name], f).offset == o, (name, f)
            lines.append(f'  printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, (size, offs) in expect.items():
        assert int(got[name]) == size
        for f, o in offs.items():
            assert int(got[f"{name}.{f}"]) == o, (name, f)
    # the Julia mirror lists the same fields in the same order with the same widths
    jl = open(os.path.join(ROOT, "julia", "BlockSparseMatricesROCm.jl")).read()
    body = re.search(r"mutable struct BsmOptions.*?\nend", jl, re.S).group(0)
    fields = re.findall(r"(\w+)::(Int32|Int64|Ptr\{Cvoid\}|NTuple\{1,Int64\})", body)
    width = {"Int32": 4, "Int64": 8, "Ptr{Cvoid}": 8, "NTuple{1,Int64}": 8}
    off = 0
    for f, t in fields:
        assert expect["bsm_options"][1][f] == off, f
        off += width[t]
    assert off == 72 and len(fields) == len(expect["bsm_options"][1])


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
