"""CPU: the C-ABI library loads without a GPU, exports every symbol the headers declare, and the
public structs have the layout of SURVEY.md Appendix B (Julia binds them by raw layout)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, hprlp

INC = os.path.join(ROOT, "include")


def declared_functions():
    names = set()
    for h in ("HPRLP.h", "batched_solver.h", "hprlp_amd.h"):
        text = open(os.path.join(INC, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = "\n".join(ln for ln in text.split("\n") if not ln.lstrip().startswith("#"))
        for mm in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text):
            names.add(mm.group(1))
    names -= {"HPRLP_DEFAULT_ARG", "defined"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    L = hprlp.lib()
    fns = declared_functions()
    assert {"create_model_from_arrays", "create_model_from_mps", "solve", "free_model", "HPRLP_main_solve",
            "solve_batched", "free_batched_results"} <= set(fns)
    missing = [f for f in fns if not hasattr(L, f)]
    assert not missing, missing
    assert L.hprlp_backend().decode() == "hip-gfx950"


def test_ctypes_mirror_sizes():
    assert C.sizeof(hprlp.CParameters) == 40
    assert C.sizeof(hprlp.CResults) == 160
    assert C.sizeof(hprlp.CBatchedResults) == 112
    assert C.sizeof(hprlp.CLPInfo) == 64
    assert C.sizeof(hprlp.CSparseMatrix) == 40
    assert hprlp.CResults.status.offset == 72 and hprlp.CResults.x.offset == 136
    assert hprlp.CParameters.CUSPARSE_spmv.offset == 32 and hprlp.CParameters.use_presolve.offset == 38
    assert hprlp.CBatchedResults.status.offset == 72 and hprlp.CBatchedResults.time.offset == 80


PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "HPRLP.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu\n", sizeof(HPRLP_parameters), sizeof(HPRLP_results), sizeof(HPRLP_batched_results),
           sizeof(LP_info_cpu), sizeof(sparseMatrix));
    printf("%zu %zu %zu %zu\n", offsetof(HPRLP_parameters, stop_tol), offsetof(HPRLP_parameters, check_iter),
           offsetof(HPRLP_results, status), offsetof(HPRLP_batched_results, iter));
    HPRLP_parameters p = HPRLP_PARAMETERS_DEFAULT;
    printf("%d %g %g %d %d %d %d\n", p.max_iter, p.stop_tol, p.time_limit, p.check_iter, (int)p.use_CR_scaling,
           (int)p.use_presolve, (int)p.CUSPARSE_spmv);
    return 0;
}
"""


@pytest.mark.parametrize("cc,std", [("gcc", "-std=c11"), ("g++", "-std=c++11")])
def test_headers_compile_without_hip_and_keep_layout(cc, std):
    """The public headers must compile with a plain host compiler, as C and as C++11 (pybind/MEX/Julia
    builds of the reference bindings use nothing else) and give the Appendix-B layout."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.c" if cc == "gcc" else "probe.cpp")
        open(src, "w").write(PROBE if cc == "gcc" else PROBE.replace("HPRLP_parameters p = HPRLP_PARAMETERS_DEFAULT;", "HPRLP_parameters p;"))
        exe = os.path.join(d, "probe")
        subprocess.check_call([cc, std, "-I", INC, src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    assert out[0].split() == ["40", "160", "112", "64", "40"]
    assert out[1].split() == ["8", "28", "72", "64"]
    assert out[2].split() == ["2147483647", "0.0001", "3600", "150", "1", "1", "0"]


def test_model_from_arrays_host_behaviour(model_mps_arrays):
    a = model_mps_arrays
    m = hprlp.Model.from_csr(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"])
    assert (m.m, m.n, m.obj_constant) == (2, 2, 0.0)
    rp, ci, v = m.csr()
    assert list(rp) == [0, 2, 4] and list(ci) == [0, 1, 0, 1] and list(v) == [1, 2, 3, 1]
    vec = m.vectors()
    assert np.isinf(vec["AL"]).all() and list(vec["AU"]) == [10, 12] and np.isinf(vec["u"]).all()
    m.free()
    # CSC input of the same matrix (column pointers) gives the same stored CSR (reference HPRLP.cu:354-396)
    m2 = hprlp.Model.from_csr(2, 2, [0, 2, 4], [0, 1, 0, 1], [1.0, 3.0, 2.0, 1.0], a["AL"], a["AU"], a["l"], a["u"], a["c"], is_csc=True)
    rp, ci, v = m2.csr()
    assert list(rp) == [0, 2, 4] and list(ci) == [0, 1, 0, 1] and list(v) == [1, 2, 3, 1]
    m2.free()


@pytest.mark.parametrize("bad", ["m0", "nnz0", "rowptr0", "rowptr_end", "null", "col_range", "decreasing"])
def test_model_from_arrays_rejects_bad_input(bad, model_mps_arrays, capfd):
    a = dict(model_mps_arrays)
    L = hprlp.lib()
    rp = np.array(a["rowptr"], np.int32); ci = np.array(a["colind"], np.int32); v = np.array(a["values"])
    AL = np.array(a["AL"]); AU = np.array(a["AU"]); l = np.array(a["l"]); u = np.array(a["u"]); c = np.array(a["c"])
    m, n, nnz = 2, 2, 4
    P = lambda x: x.ctypes.data_as(hprlp.c_dbl_p)
    I = lambda x: x.ctypes.data_as(hprlp.c_int_p)
    pc = P(c)
    if bad == "m0": m = 0
    if bad == "nnz0": nnz = 0
    if bad == "rowptr0": rp[0] = 1
    if bad == "rowptr_end": rp[2] = 3
    if bad == "null": pc = None
    if bad == "col_range": ci[3] = 2
    if bad == "decreasing": rp[1] = 5
    ptr = L.create_model_from_arrays(m, n, nnz, I(rp), I(ci), P(v), P(AL), P(AU), P(l), P(u), pc, False)
    assert not ptr                                   # NULL + message on stderr (reference HPRLP.cu:329-337)
    assert "[error]" in capfd.readouterr().err
    L.free_model(None)                               # NULL is a no-op (reference HPRLP.cu:530)


def test_gpu_calls_fail_loudly_without_gpu(model_mps_arrays):
    """No CPU fallback exists: on a box without a GPU the solve returns status ERROR, it does not
    silently compute on the host."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    a = model_mps_arrays
    m = hprlp.Model.from_csr(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"])
    r = m.solve(hprlp.Parameters(use_presolve=False))
    assert r.status == "ERROR" and r.x is None
    with pytest.raises(RuntimeError):
        hprlp.Solver(m)
    m.free()


def test_warmup_without_a_gpu_says_so():
    """hprlp_warmup (include/hprlp_amd.h, round 4) on a host without a GPU: -1 and a message, no crash; the model API stays usable."""
    if os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK):
        pytest.skip("this host has a GPU")
    L = hprlp.lib()
    L.hprlp_warmup.restype = C.c_int
    assert L.hprlp_warmup(0) == -1
    assert "GPU" in hprlp.last_error()


def test_banded_generator_is_row_consistent():
    import bench_helpers as bh
    rp, ci, v = bh.gen_banded(5000, 5000, 20, 100, seed=7)
    rp2, ci2, v2 = bh.gen_banded(5000, 5000, 20, 100, seed=7, row0=1200, rows=800)
    assert np.array_equal(ci[1200 * 20:2000 * 20], ci2) and np.array_equal(v[1200 * 20:2000 * 20], v2)
    c = ci.reshape(5000, 20)
    assert (np.diff(c, axis=1) > 0).all() and c.min() >= 0 and c.max() < 5000
    near = np.abs(c - np.arange(5000)[:, None]) <= 121
    assert 0.90 < near.mean() < 0.99


@pytest.mark.parametrize("bad", ["col_range", "col_negative", "decreasing", "rowptr_end", "dims"])
def test_solve_with_presolve_refuses_a_corrupt_hand_built_model(bad, model_mps_arrays, capfd):
    """solve() presolves by default, on the host, BEFORE anything is uploaded: a caller-filled LP_info_cpu with a bad
    index must come back as status ERROR, not as an out-of-bounds host access (create_model_from_arrays validates;
    a hand-built struct -- what the Julia/MATLAB bindings could pass -- does not go through it)."""
    a = model_mps_arrays
    rp = np.array(a["rowptr"], np.int32); ci = np.array(a["colind"], np.int32); v = np.array(a["values"])
    vec = {k: np.array(a[k], np.float64) for k in ("AL", "AU", "l", "u", "c")}
    if bad == "col_range": ci[3] = 1 << 20
    if bad == "col_negative": ci[0] = -7
    if bad == "decreasing": rp[1] = 5
    if bad == "rowptr_end": rp[2] = 3
    A = hprlp.CSparseMatrix(row=2, col=3 if bad == "dims" else 2, numElements=4,
                            colIndex=ci.ctypes.data_as(hprlp.c_int_p), rowPtr=rp.ctypes.data_as(hprlp.c_int_p),
                            value=v.ctypes.data_as(hprlp.c_dbl_p))
    lp = hprlp.CLPInfo(m=2, n=2, A=C.pointer(A), obj_constant=0.0,
                       **{k: x.ctypes.data_as(hprlp.c_dbl_p) for k, x in vec.items()})
    L = hprlp.lib()
    prm = hprlp.Parameters(use_presolve=True).to_c()
    L.solve.restype = hprlp.CResults
    r = L.solve(C.byref(lp), C.byref(prm))
    assert r.status.decode() == "ERROR" and not r.x and not r.y and not r.z
    assert "invalid model" in capfd.readouterr().err


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("src", ["examples/c/example_direct_lp.c", "examples/c/example_mps_file.c",
                                 "examples/c/example_batched_lp.c", "examples/cpp/example_direct_lp.cpp",
                                 "examples/cpp/example_mps_file.cpp", "src/solve_mps_file.cpp"])
def test_reference_examples_compile_and_link_unchanged(src, tmp_path):
    """Drop-in boundary: every example of the reference and its CLI driver must compile against include/ and link
    against lib/libhprlp.so as they are (the reference builds the "C" ones as C++ too, examples/c/Makefile:45).
    example_direct_lp.cpp uses INFINITY with only <iostream>/<iomanip>/HPRLP.h included."""
    path = os.path.join(REF, src)
    if not os.path.exists(path):
        pytest.skip(f"{src} not in this reference checkout")
    exe = str(tmp_path / "a.out")
    cmd = ["g++", "-std=c++11", "-x", "c++", "-I" + INC, path, "-o", exe, "-L" + os.path.join(ROOT, "lib"), "-lhprlp",
           "-Wl,-rpath," + os.path.join(ROOT, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # the symbols it binds are ours
    nm = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert re.search(r"\b(solve|create_model_from_arrays|create_model_from_mps|solve_batched)\b", nm)


def test_device_block_cache_is_keyed_by_the_owning_device():
    """csrc/alloc.cpp: freed device blocks are filed under the device that owns them (not the freeing thread's current
    device) and only handed to requests of that device; the cap is per device.  Pure bookkeeping, runs without a GPU."""
    L = hprlp.lib()
    L.hprlp_alloc_cache_selftest.restype = C.c_int
    assert L.hprlp_alloc_cache_selftest() == 0


def test_environment_switches_all_go_through_one_table():
    """csrc/env.h (round 5): every HPRLP_* variable the library reads is an entry of env.cpp's table -- at most ten
    integrator-facing ones, always honoured; the rest test hooks, honoured only under HPRLP_TEST_HOOKS=1 -- and no source
    reads the environment past it.  hprlp_env_switches() lists the table; INTEGRATION.md carries the same names."""
    import ctypes as C
    import glob
    import re
    L = hprlp.lib()
    n = L.hprlp_env_switches(None, 0)
    buf = C.create_string_buffer(n + 1)
    assert L.hprlp_env_switches(buf, n + 1) == n
    rows = [ln.split("\t") for ln in buf.value.decode().splitlines()]
    assert all(len(r) == 3 and r[0].startswith("HPRLP_") and r[1] in ("integrator", "hook") and r[2] for r in rows), rows[:3]
    table = {r[0]: r[1] for r in rows}
    assert len(table) == len(rows)
    assert sum(1 for k in table.values() if k == "integrator") <= 10
    assert table["HPRLP_TEST_HOOKS"] == "integrator" and table["HPRLP_TIMING"] == "integrator" and table["HPRLP_TILE_ROWS"] == "hook"
    used = set()
    for f in glob.glob(os.path.join(ROOT, "hpr-lp-c_amd", "csrc", "*")):
        src = open(f).read()
        if not f.endswith("env.cpp"):
            assert not re.search(r'getenv\s*\(', src), f"{f} reads the environment past env_get()"
        used |= set(re.findall(r'env_(?:get|on)\("(HPRLP_[A-Z0-9_]+)"\)', src))
    assert used and used <= set(table), used - set(table)
    assert set(table) - used <= {"HPRLP_TEST_HOOKS"}, set(table) - used          # no dead entries
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [k for k in table if k not in doc]
    assert not missing, missing
