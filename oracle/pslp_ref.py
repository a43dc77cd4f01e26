"""ctypes loader for oracle/_ref/libpslp_ref.so: the presolver the reference vendors (PSLP v0.0.8), built
unchanged from /root/reference/third_party/PSLP by `make -C oracle refpslp`.  TEST INFRASTRUCTURE ONLY -- the
checker for our own presolve (hpr-lp-c_amd/csrc/presolve.cpp); the product never loads it.
Interface followed: third_party/PSLP/include/PSLP/PSLP_API.h:42-134 and the reference's call sequence in
src/pslp_integration.cpp:223-330 (default_settings, verbose off, new_presolver, run_presolver, postsolve)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libpslp_ref.so")


def available():
    return os.path.exists(_PATH)


class Settings(C.Structure):  # PSLP_API.h:42-56
    _fields_ = [("ston_cols", C.c_bool), ("dton_eq", C.c_bool), ("parallel_rows", C.c_bool), ("parallel_cols", C.c_bool),
                ("primal_propagation", C.c_bool), ("finite_bound_tightening", C.c_bool), ("dual_fix", C.c_bool),
                ("relax_bounds", C.c_bool), ("max_shift", C.c_int), ("max_time", C.c_double), ("verbose", C.c_bool)]


class PresolvedProblem(C.Structure):  # PSLP_API.h:59-82
    _fields_ = [("Ax", C.POINTER(C.c_double)), ("Ai", C.POINTER(C.c_int)), ("Ap", C.POINTER(C.c_int)),
                ("m", C.c_size_t), ("n", C.c_size_t), ("nnz", C.c_size_t),
                ("lhs", C.POINTER(C.c_double)), ("rhs", C.POINTER(C.c_double)), ("c", C.POINTER(C.c_double)),
                ("lbs", C.POINTER(C.c_double)), ("ubs", C.POINTER(C.c_double)), ("obj_offset", C.c_double)]


class Solution(C.Structure):  # PSLP_sol.h:29-36
    _fields_ = [("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)), ("z", C.POINTER(C.c_double)),
                ("dim_x", C.c_size_t), ("dim_y", C.c_size_t)]


class Presolver(C.Structure):  # PSLP_API.h:95-102
    _fields_ = [("stats", C.c_void_p), ("stgs", C.POINTER(Settings)), ("prob", C.c_void_p),
                ("reduced_prob", C.POINTER(PresolvedProblem)), ("sol", C.POINTER(Solution))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        L.default_settings.restype = C.POINTER(Settings)
        L.free_settings.argtypes = [C.POINTER(Settings)]
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.new_presolver.restype = C.POINTER(Presolver)
        L.new_presolver.argtypes = [dp, ip, ip, C.c_size_t, C.c_size_t, C.c_size_t, dp, dp, dp, dp, dp, C.POINTER(Settings)]
        L.run_presolver.restype = C.c_uint8
        L.run_presolver.argtypes = [C.POINTER(Presolver)]
        L.postsolve.argtypes = [C.POINTER(Presolver), dp, dp, dp]
        L.free_presolver.argtypes = [C.POINTER(Presolver)]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class RefPresolve:
    """Run the reference's presolver on lhs <= A x <= rhs, lb <= x <= ub, min c.x (CSR input)."""

    def __init__(self, m, n, rowptr, colind, values, AL, AU, l, u, c):
        L = lib()
        self.m, self.n = m, n
        self._keep = [np.ascontiguousarray(a, dtype=t) for a, t in
                      ((values, np.float64), (colind, np.int32), (rowptr, np.int32), (AL, np.float64), (AU, np.float64),
                       (l, np.float64), (u, np.float64), (c, np.float64))]
        v, ci, rp, al, au, lo, up, cc = self._keep
        self.stgs = L.default_settings()
        self.stgs.contents.verbose = False
        self.p = L.new_presolver(_d(v), _i(ci), _i(rp), m, n, len(v), _d(al), _d(au), _d(lo), _d(up), _d(cc), self.stgs)
        if not self.p:
            raise RuntimeError("new_presolver failed")
        self.status = int(L.run_presolver(self.p))
        r = self.p.contents.reduced_prob.contents
        self.rm, self.rn, self.rnnz = int(r.m), int(r.n), int(r.nnz)
        self.obj_offset = float(r.obj_offset)
        g = lambda ptr, k, t: np.ctypeslib.as_array(ptr, shape=(k,)).astype(t).copy() if k > 0 else np.zeros(0, dtype=t)
        self.Ap = g(r.Ap, self.rm + 1, np.int32) if self.rm >= 0 else np.zeros(1, np.int32)
        self.Ai = g(r.Ai, self.rnnz, np.int32)
        self.Ax = g(r.Ax, self.rnnz, np.float64)
        self.lhs, self.rhs = g(r.lhs, self.rm, np.float64), g(r.rhs, self.rm, np.float64)
        self.lbs, self.ubs, self.c = g(r.lbs, self.rn, np.float64), g(r.ubs, self.rn, np.float64), g(r.c, self.rn, np.float64)

    def postsolve(self, x, y, z):
        L = lib()
        x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
        L.postsolve(self.p, _d(x), _d(y), _d(z))
        s = self.p.contents.sol.contents
        return (np.ctypeslib.as_array(s.x, shape=(int(s.dim_x),)).copy(),
                np.ctypeslib.as_array(s.y, shape=(int(s.dim_y),)).copy(),
                np.ctypeslib.as_array(s.z, shape=(int(s.dim_x),)).copy())

    def close(self):
        if self.p:
            lib().free_presolver(self.p)
            lib().free_settings(self.stgs)
            self.p = None
