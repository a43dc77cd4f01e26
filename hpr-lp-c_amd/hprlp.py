"""ctypes host mirror of the HPR-LP boundary (lib/libhprlp.so, include/HPRLP.h + hprlp_amd.h).

Same names and argument meaning as the reference's Python package (reference
bindings/python/hprlp/{model,parameters,results,solver}.py): Parameters, Model.from_arrays /
Model.from_mps, Model.solve, solve_batched, Results.  Plumbing only: every number is produced by the
HIP library; nothing here falls back to a CPU path, and loading fails loudly when the library (or,
for solves, a GPU) is missing.
"""
import ctypes as C
import os
import time
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("HPRLP_LIB") or os.path.join(_ROOT, "lib", "libhprlp.so")  # HPRLP_LIB: developer builds (kernel shape variants)

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class CParameters(C.Structure):  # include/structs.h HPRLP_parameters (40 bytes)
    _fields_ = [
        ("max_iter", C.c_int), ("stop_tol", C.c_double), ("time_limit", C.c_double),
        ("device_number", C.c_int), ("check_iter", C.c_int),
        ("CUSPARSE_spmv", C.c_bool), ("autotune_verbose", C.c_bool), ("use_CR_scaling", C.c_bool),
        ("use_Ruiz_scaling", C.c_bool), ("use_Pock_Chambolle_scaling", C.c_bool),
        ("use_bc_scaling", C.c_bool), ("use_presolve", C.c_bool),
    ]


class CResults(C.Structure):  # HPRLP_results (160 bytes)
    _fields_ = [
        ("residuals", C.c_double), ("primal_obj", C.c_double), ("gap", C.c_double),
        ("time4", C.c_double), ("time6", C.c_double), ("time8", C.c_double), ("time", C.c_double),
        ("iter4", C.c_int), ("iter6", C.c_int), ("iter8", C.c_int), ("iter", C.c_int),
        ("status", C.c_char * 64),
        ("x", c_dbl_p), ("y", c_dbl_p), ("z", c_dbl_p),
    ]


class CBatchedResults(C.Structure):  # HPRLP_batched_results (112 bytes)
    _fields_ = [
        ("m", C.c_int), ("n", C.c_int), ("batch_size", C.c_int),
        ("x", c_dbl_p), ("y", c_dbl_p), ("z", c_dbl_p),
        ("primal_obj", c_dbl_p), ("residuals", c_dbl_p), ("gap", c_dbl_p),
        ("iter", c_int_p), ("status", C.POINTER(C.c_char)),
        ("time", C.c_double), ("setup_time", C.c_double), ("solve_time", C.c_double), ("power_time", C.c_double),
    ]


class CSparseMatrix(C.Structure):
    _fields_ = [("row", C.c_int), ("col", C.c_int), ("numElements", C.c_int),
                ("colIndex", c_int_p), ("rowPtr", c_int_p), ("value", c_dbl_p)]


class CLPInfo(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("A", C.POINTER(CSparseMatrix)),
                ("AL", c_dbl_p), ("AU", c_dbl_p), ("c", c_dbl_p), ("l", c_dbl_p), ("u", c_dbl_p),
                ("obj_constant", C.c_double)]


class CShard(C.Structure):
    """hprlp_shard (include/hprlp_amd.h): one rank's rows of A and of A^T with its vector slices."""
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("row_off", C.c_int), ("m_loc", C.c_int), ("col_off", C.c_int),
                ("n_loc", C.c_int), ("A_rowptr", c_int_p), ("A_col", c_int_p), ("A_val", c_dbl_p),
                ("AT_rowptr", c_int_p), ("AT_col", c_int_p), ("AT_val", c_dbl_p),
                ("AL", c_dbl_p), ("AU", c_dbl_p), ("l", c_dbl_p), ("u", c_dbl_p), ("c", c_dbl_p), ("obj_constant", C.c_double)]


class CTraceRow(C.Structure):
    _fields_ = [("iter", C.c_int), ("restart_flag", C.c_int)] + [
        (k, C.c_double)
        for k in ("err_Rp", "err_Rd", "primal_obj", "dual_obj", "gap", "kkt", "sigma", "current_gap", "lambda_max")
    ]


_lib = None
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def lib():
    """Load lib/libhprlp.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    L.create_model_from_arrays.restype = C.POINTER(CLPInfo)
    L.create_model_from_arrays.argtypes = [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p,
                                           c_dbl_p, c_dbl_p, c_dbl_p, C.c_bool]
    L.create_model_from_mps.restype = C.POINTER(CLPInfo)
    L.create_model_from_mps.argtypes = [C.c_char_p]
    L.free_model.argtypes = [C.POINTER(CLPInfo)]
    L.solve.restype = CResults
    L.solve.argtypes = [C.POINTER(CLPInfo), C.POINTER(CParameters)]
    L.HPRLP_main_solve.restype = CResults
    L.HPRLP_main_solve.argtypes = [C.POINTER(CLPInfo), C.POINTER(CParameters)]
    L.solve_batched.restype = CBatchedResults
    L.solve_batched.argtypes = [C.POINTER(CLPInfo), C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p,
                                C.POINTER(CParameters)]
    L.free_batched_results.argtypes = [C.POINTER(CBatchedResults)]
    L.hprlp_last_error.restype = C.c_char_p
    L.hprlp_backend.restype = C.c_char_p
    L.hprlp_solver_create.restype = C.c_void_p
    L.hprlp_solver_create.argtypes = [C.POINTER(CLPInfo), C.POINTER(CParameters)]
    L.hprlp_solver_destroy.argtypes = [C.c_void_p]
    L.hprlp_solver_set_verbose.argtypes = [C.c_void_p, C.c_int]
    L.hprlp_solver_scale.argtypes = [C.c_void_p]
    L.hprlp_solver_power_iteration.restype = C.c_double
    L.hprlp_solver_power_iteration.argtypes = [C.c_void_p, C.c_int, C.c_double, c_int_p]
    L.hprlp_solver_init.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.hprlp_solver_iterate.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.hprlp_solver_residuals.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dbl_p]
    L.hprlp_solver_restart.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p]
    L.hprlp_solver_weighted_norm.restype = C.c_double
    L.hprlp_solver_weighted_norm.argtypes = [C.c_void_p]
    L.hprlp_solver_run.argtypes = [C.c_void_p, C.POINTER(CResults), C.POINTER(CTraceRow), C.c_int, c_int_p]
    L.hprlp_solver_get_vector.restype = C.c_long
    L.hprlp_solver_get_vector.argtypes = [C.c_void_p, C.c_char_p, c_dbl_p, C.c_long]
    L.hprlp_solver_set_vector.argtypes = [C.c_void_p, C.c_char_p, c_dbl_p, C.c_long]
    L.hprlp_solver_get_scalars.argtypes = [C.c_void_p, c_dbl_p]
    L.hprlp_solver_info.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
    L.hprlp_solver_time_iterations.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]
    L.hprlp_presolve_run.restype = C.c_void_p
    L.hprlp_presolve_run.argtypes = [C.POINTER(CLPInfo)]
    L.hprlp_presolve_reduced.restype = C.POINTER(CLPInfo)
    L.hprlp_presolve_reduced.argtypes = [C.c_void_p]
    L.hprlp_presolve_stats.argtypes = [C.c_void_p, c_int_p]
    L.hprlp_presolve_postsolve.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]
    L.hprlp_presolve_free.argtypes = [C.c_void_p]
    L.hprlp_original_kkt.argtypes = [C.POINTER(CLPInfo), c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]
    _lib = L
    return L


def last_error():
    return lib().hprlp_last_error().decode()


class Parameters:
    """Solver parameters; defaults of reference include/structs.h:26-39."""

    _names = [f for f, _ in CParameters._fields_]

    def __init__(self, **kw):
        self.max_iter = 2**31 - 1
        self.stop_tol = 1e-4
        self.time_limit = 3600.0
        self.device_number = 0
        self.check_iter = 150
        self.CUSPARSE_spmv = False
        self.autotune_verbose = False
        self.use_CR_scaling = True
        self.use_Ruiz_scaling = True
        self.use_Pock_Chambolle_scaling = True
        self.use_bc_scaling = True
        self.use_presolve = True
        for k, v in kw.items():
            if k not in self._names:
                raise AttributeError(k)
            setattr(self, k, v)

    def to_c(self):
        return CParameters(*[getattr(self, k) for k in self._names])


def _as(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _take(ptr, n):
    """A malloc'd result array as a numpy array (the caller owns HPRLP_results.x/y/z).  Short vectors are copied and freed at
    once; long ones are adopted without a copy -- the array keeps the C buffer and free()s it when the last view is gone (three
    copies of 80 MB were 36 ms of a config-5 solve's wall time on the Python side)."""
    if not ptr:
        return None
    view = np.ctypeslib.as_array(ptr, shape=(n,))
    if n < (1 << 20):
        out = view.copy()
        _libc.free(C.cast(ptr, C.c_void_p))
        return out
    addr = C.cast(ptr, C.c_void_p).value
    buf = (C.c_double * n).from_address(addr)
    weakref.finalize(buf, _libc.free, C.c_void_p(addr))
    return np.frombuffer(buf, dtype=np.float64)  # (its .base keeps `buf` alive)


class Results:
    def __init__(self, cres, m, n):
        self.status = cres.status.decode()
        for k in ("residuals", "primal_obj", "gap", "time4", "time6", "time8", "time", "iter4", "iter6", "iter8", "iter"):
            setattr(self, k, getattr(cres, k))
        self.x = _take(cres.x, n)
        self.y = _take(cres.y, m)
        self.z = _take(cres.z, n)


class Model:
    """LP model: min c'x s.t. AL <= Ax <= AU, l <= x <= u (wraps LP_info_cpu*)."""

    def __init__(self, ptr):
        if not ptr:
            raise RuntimeError("model creation failed (see stderr)")
        self._ptr = ptr

    @property
    def m(self):
        return self._ptr.contents.m

    @property
    def n(self):
        return self._ptr.contents.n

    @property
    def obj_constant(self):
        return self._ptr.contents.obj_constant

    @property
    def nnz(self):
        return self._ptr.contents.A.contents.numElements

    @staticmethod
    def from_csr(m, n, rowptr, colind, values, AL, AU, l, u, c, is_csc=False):
        rp = _as(rowptr, np.int32); ci = _as(colind, np.int32); v = _as(values, np.float64)
        AL = _as(AL, np.float64); AU = _as(AU, np.float64); l = _as(l, np.float64); u = _as(u, np.float64)
        c = _as(c, np.float64)
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        ptr = lib().create_model_from_arrays(m, n, len(v), rp.ctypes.data_as(c_int_p), ci.ctypes.data_as(c_int_p),
                                             P(v), P(AL), P(AU), P(l), P(u), P(c), bool(is_csc))
        return Model(ptr)

    @staticmethod
    def from_arrays(A, AL, AU, l, u, c):
        """A: dense ndarray or scipy.sparse matrix (reference bindings/python/hprlp/model.py:96-174)."""
        from scipy import sparse
        A = sparse.csr_matrix(A)
        A.sort_indices()
        m, n = A.shape
        return Model.from_csr(m, n, A.indptr, A.indices, A.data, AL, AU, l, u, c)

    @staticmethod
    def from_mps(path):
        return Model(lib().create_model_from_mps(str(path).encode()))

    def csr(self):
        """Host copy of the stored CSR arrays (rowPtr, colIndex, value)."""
        A = self._ptr.contents.A.contents
        rp = np.ctypeslib.as_array(A.rowPtr, shape=(A.row + 1,)).copy()
        ci = np.ctypeslib.as_array(A.colIndex, shape=(A.numElements,)).copy()
        v = np.ctypeslib.as_array(A.value, shape=(A.numElements,)).copy()
        return rp, ci, v

    def vectors(self):
        p = self._ptr.contents
        g = lambda q, k: np.ctypeslib.as_array(q, shape=(k,)).copy()
        return dict(AL=g(p.AL, p.m), AU=g(p.AU, p.m), l=g(p.l, p.n), u=g(p.u, p.n), c=g(p.c, p.n))

    def solve(self, param=None):
        cp = (param or Parameters()).to_c()
        L = lib()
        t0 = time.perf_counter()
        res = L.solve(self._ptr, C.byref(cp))
        wall = time.perf_counter() - t0
        r = Results(res, self.m, self.n)
        r.c_call_wall_s = wall  # the caller's clock around the C call alone (before the solution vectors are wrapped)
        return r

    def free(self):
        if self._ptr:
            lib().free_model(self._ptr)
            self._ptr = None


class Presolved:
    """Host-side presolve of a model (include/hprlp_amd.h hprlp_presolve_*): `reduced` is a Model view owned by
    this object; postsolve() maps a reduced primal-dual solution back.  Raises if the model is left unchanged."""

    def __init__(self, model):
        self.model = model
        self.h = lib().hprlp_presolve_run(model._ptr)
        if not self.h:
            raise RuntimeError(last_error())
        self.reduced = Model(lib().hprlp_presolve_reduced(self.h))
        out = (C.c_int * 16)()
        lib().hprlp_presolve_stats(self.h, out)
        keys = ("m", "n", "fixed_cols", "empty_cols", "singleton_rows", "empty_rows", "redundant_rows", "passes", "dual_fixed_cols", "slack_cols", "parallel_rows", "parallel_cols", "forcing_rows", "doubleton_rows", "tightened_bounds", "rounds")
        self.stats = dict(zip(keys, [int(v) for v in out]))

    def postsolve(self, xr, yr, zr):
        xr, yr, zr = _as(xr, np.float64), _as(yr, np.float64), _as(zr, np.float64)
        x, y, z = np.zeros(self.model.n), np.zeros(self.model.m), np.zeros(self.model.n)
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        if lib().hprlp_presolve_postsolve(self.h, P(xr), P(yr), P(zr), P(x), P(y), P(z)) != 0:
            raise RuntimeError("postsolve failed")
        return x, y, z

    def free(self):
        if self.h:
            self.reduced._ptr = None  # owned by the presolve object
            lib().hprlp_presolve_free(self.h)
            self.h = None


def last_solve_phases():
    """Phases [s] of this thread's last HPRLP_main_solve (hprlp_last_solve_phases)."""
    out = (C.c_double * 8)()
    L = lib()
    L.hprlp_last_solve_phases.argtypes = [C.POINTER(C.c_double)]
    L.hprlp_last_solve_phases(out)
    keys = ("device_setup", "scaling", "power_iteration", "loop", "collect_solution", "teardown", "whole_call")
    return dict(zip(keys, [float(v) for v in out]))


def original_kkt(model, x, y, z):
    """Relative primal / dual infeasibility and gap of (x, y, z) on the model as given."""
    x, y, z = _as(x, np.float64), _as(y, np.float64), _as(z, np.float64)
    out = np.zeros(5)
    P = lambda a: a.ctypes.data_as(c_dbl_p)
    if lib().hprlp_original_kkt(model._ptr, P(x), P(y), P(z), P(out)) != 0:
        raise RuntimeError("original_kkt failed")
    return dict(primal_feas=out[0], dual_feas=out[1], gap=out[2], primal_obj=out[3], dual_obj=out[4])


def solve(A, AL, AU, l, u, c, param=None):
    model = Model.from_arrays(A, AL, AU, l, u, c)
    try:
        return model.solve(param)
    finally:
        model.free()


def solve_batched(model, Cmat, AL, AU, l, u, obj_constants=None, param=None):
    """Cmat,l,u: (n,B) arrays; AL,AU: (m,B) arrays (any layout; passed column-major as the ABI asks)."""
    Cmat = np.asfortranarray(Cmat, dtype=np.float64)
    B = Cmat.shape[1]
    F = lambda a: np.asfortranarray(a, dtype=np.float64)
    AL, AU, l, u = F(AL), F(AU), F(l), F(u)
    P = lambda a: a.ctypes.data_as(c_dbl_p)
    oc = None if obj_constants is None else _as(obj_constants, np.float64)
    cp = (param or Parameters()).to_c()
    res = lib().solve_batched(model._ptr, B, P(Cmat), P(AL), P(AU), P(l), P(u), None if oc is None else P(oc),
                              C.byref(cp))
    m, n = model.m, model.n
    g = lambda q, k: None if not q else np.ctypeslib.as_array(q, shape=(k,)).copy()
    out = dict(batch_size=res.batch_size, time=res.time, setup_time=res.setup_time, solve_time=res.solve_time,
               power_time=res.power_time)
    out["x"] = None if not res.x else g(res.x, n * B).reshape(B, n).T
    out["y"] = None if not res.y else g(res.y, m * B).reshape(B, m).T
    out["z"] = None if not res.z else g(res.z, n * B).reshape(B, n).T
    out["primal_obj"] = g(res.primal_obj, B); out["residuals"] = g(res.residuals, B); out["gap"] = g(res.gap, B)
    out["iter"] = None if not res.iter else np.ctypeslib.as_array(res.iter, shape=(B,)).copy()
    raw = C.string_at(res.status, 64 * res.batch_size) if res.status else b""
    out["status"] = [raw[64 * k:64 * (k + 1)].split(b"\0")[0].decode() for k in range(res.batch_size if raw else 0)]
    lib().free_batched_results(C.byref(res))
    return out


class Solver:
    """Step-level handle (include/hprlp_amd.h) used by the parity tests and bench.py."""

    def __init__(self, model, param=None):
        cp = (param or Parameters()).to_c()
        self.model = model
        self.h = lib().hprlp_solver_create(model._ptr, C.byref(cp))
        if not self.h:
            raise RuntimeError("hprlp_solver_create failed: " + last_error())

    @classmethod
    def create_dist(cls, model, param, rank, size, unique_id=None):
        """One rank of the row-partitioned solve (hprlp_solver_create_dist).  unique_id: 128-byte numpy
        uint8 array from dist_unique_id() (rank 0) broadcast to every rank."""
        L = lib()
        L.hprlp_solver_create_dist.restype = C.c_void_p
        L.hprlp_solver_create_dist.argtypes = [C.POINTER(CLPInfo), C.POINTER(CParameters), C.c_int, C.c_int,
                                               C.c_void_p, C.c_int]
        self = cls.__new__(cls)
        self.model = model
        cp = (param or Parameters()).to_c()
        uid = None if unique_id is None else unique_id.ctypes.data_as(C.c_void_p)
        self.h = L.hprlp_solver_create_dist(model._ptr, C.byref(cp), rank, size, uid, 0 if unique_id is None else len(unique_id))
        if not self.h:
            raise RuntimeError("hprlp_solver_create_dist failed: " + last_error())
        self._set_local_sizes(rank, size)
        return self

    @classmethod
    def create_dist_from_shard(cls, shard, param, rank, size, unique_id=None, group=None):
        """One rank of the row-partitioned solve from a shard assembled by the caller (shard.ShardArrays): no rank holds the
        whole matrix.  group: a local_group() handle runs the ranks as threads of this process (tests) instead of RCCL."""
        L = lib()
        self = cls.__new__(cls)
        self.model = shard            # has .m / .n; keeps the arrays alive until the solver is created
        cp = (param or Parameters()).to_c()
        if group is not None:
            L.hprlp_solver_create_local_from_shard.restype = C.c_void_p
            L.hprlp_solver_create_local_from_shard.argtypes = [C.POINTER(CShard), C.POINTER(CParameters), C.c_int, C.c_int, C.c_void_p]
            self.h = L.hprlp_solver_create_local_from_shard(C.byref(shard.c_shard), C.byref(cp), rank, size, group)
        else:
            L.hprlp_solver_create_dist_from_shard.restype = C.c_void_p
            L.hprlp_solver_create_dist_from_shard.argtypes = [C.POINTER(CShard), C.POINTER(CParameters), C.c_int, C.c_int,
                                                              C.c_void_p, C.c_int]
            uid = None if unique_id is None else unique_id.ctypes.data_as(C.c_void_p)
            self.h = L.hprlp_solver_create_dist_from_shard(C.byref(shard.c_shard), C.byref(cp), rank, size, uid,
                                                           0 if unique_id is None else len(unique_id))
        if not self.h:
            raise RuntimeError("hprlp_solver_create_dist_from_shard failed: " + last_error())
        self.row_off, self.m_loc, self.col_off, self.n_loc = shard.row_off, shard.m_loc, shard.col_off, shard.n_loc
        return self

    def _set_local_sizes(self, rank, size):
        """run()/get() of a sharded solver return this rank's slices: rows [row_off, row_off+m_loc), ..."""
        off, cnt = C.c_int(), C.c_int()
        lib().hprlp_partition(self.model.m, size, rank, C.byref(off), C.byref(cnt))
        self.row_off, self.m_loc = off.value, cnt.value
        lib().hprlp_partition(self.model.n, size, rank, C.byref(off), C.byref(cnt))
        self.col_off, self.n_loc = off.value, cnt.value

    @classmethod
    def create_local(cls, model, param, rank, size, group):
        """Rank `rank` of a `size`-rank solve whose ranks are host threads of this process (hprlp_solver_create_local);
        `group` comes from local_group(size).  Call from the rank's own thread."""
        L = lib()
        L.hprlp_solver_create_local.restype = C.c_void_p
        L.hprlp_solver_create_local.argtypes = [C.POINTER(CLPInfo), C.POINTER(CParameters), C.c_int, C.c_int, C.c_void_p]
        self = cls.__new__(cls)
        self.model = model
        cp = (param or Parameters()).to_c()
        self.h = L.hprlp_solver_create_local(model._ptr, C.byref(cp), rank, size, group)
        if not self.h:
            raise RuntimeError("hprlp_solver_create_local failed: " + last_error())
        self._set_local_sizes(rank, size)
        return self

    @staticmethod
    def local_group(size):
        L = lib()
        L.hprlp_local_group_create.restype = C.c_void_p
        L.hprlp_local_group_create.argtypes = [C.c_int]
        g = L.hprlp_local_group_create(size)
        if not g:
            raise RuntimeError(last_error())
        return C.c_void_p(g)

    @staticmethod
    def free_local_group(group):
        L = lib()
        L.hprlp_local_group_destroy.argtypes = [C.c_void_p]
        L.hprlp_local_group_destroy(group)

    def dist_loopback(self, count=100000):
        L = lib()
        L.hprlp_solver_dist_loopback.argtypes = [C.c_void_p, C.c_int]
        self._chk(L.hprlp_solver_dist_loopback(self.h, int(count)))

    def dist_info(self):
        L = lib()
        L.hprlp_solver_dist_info.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        out = (C.c_long * 8)()
        self._chk(L.hprlp_solver_dist_info(self.h, out))
        keys = ("m_sparse", "m_sent", "m_received", "n_sparse", "n_sent", "n_received", "m_requests", "n_requests")
        return dict(zip(keys, [int(v) for v in out]))

    def dist_comm_info(self):
        """What the transport reports (RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice) for the main communicator
        and for the exchange stream's own one (ranks 0 if there is none)."""
        L = lib()
        L.hprlp_solver_dist_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        out = (C.c_long * 8)()
        self._chk(L.hprlp_solver_dist_comm_info(self.h, out))
        keys = ("comm_ranks", "comm_rank", "comm_device", "xcomm_ranks", "xcomm_rank", "xcomm_device", "hip_device", "overlap")
        return dict(zip(keys, [int(v) for v in out]))

    @staticmethod
    def dist_unique_id(ids=1):
        """ids=2: a second id for the exchange stream's own communicator (one communicator, one stream)."""
        uid = np.zeros(128 * ids, np.uint8)
        if lib().hprlp_dist_unique_id(uid.ctypes.data_as(C.c_void_p), 128 * ids) != 0:
            raise RuntimeError(last_error())
        return uid

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(last_error())
        return rc

    def close(self):
        if self.h:
            lib().hprlp_solver_destroy(self.h)
            self.h = None

    def scale(self):
        self._chk(lib().hprlp_solver_scale(self.h))

    def power_iteration(self, max_iter=5000, tol=1e-4):
        it = C.c_int(0)
        lam = lib().hprlp_solver_power_iteration(self.h, max_iter, tol, C.byref(it))
        if lam < 0:
            raise RuntimeError(last_error())
        return lam, it.value

    def init(self, sigma=-1.0, lambda_max=1.0):
        self._chk(lib().hprlp_solver_init(self.h, sigma, lambda_max))

    def reset(self):
        """All iterates back to zero (hprlp_solver_reset_iterates); follow with init()."""
        L = lib()
        L.hprlp_solver_reset_iterates.argtypes = [C.c_void_p]
        L.hprlp_solver_reset_iterates.restype = C.c_int
        self._chk(L.hprlp_solver_reset_iterates(self.h))

    def iterate(self, normal, then_check=False):
        self._chk(lib().hprlp_solver_iterate(self.h, int(normal), int(bool(then_check))))

    def residuals(self, it, compute_gap=False):
        out = np.zeros(8)
        self._chk(lib().hprlp_solver_residuals(self.h, int(it), int(bool(compute_gap)), out.ctypes.data_as(c_dbl_p)))
        return dict(zip(("err_Rp", "err_Rd", "primal_obj", "dual_obj", "gap", "kkt", "weighted_norm", "lambda_max"), out))

    def restart(self, current_gap, best_gap, best_sigma, err_Rd, err_Rp, rel_gap):
        a = np.array([current_gap, best_gap, best_sigma, err_Rd, err_Rp, rel_gap], dtype=np.float64)
        s = C.c_double(0)
        self._chk(lib().hprlp_solver_restart(self.h, a.ctypes.data_as(c_dbl_p), C.byref(s)))
        return s.value

    def weighted_norm(self):
        return lib().hprlp_solver_weighted_norm(self.h)

    def get(self, name):
        info = self.info()
        cap = max(info["m"], info["n"], info["nnz"], 1)
        buf = np.zeros(cap)
        k = lib().hprlp_solver_get_vector(self.h, name.encode(), buf.ctypes.data_as(c_dbl_p), cap)
        if k < 0:
            raise RuntimeError(last_error())
        return buf[:k].copy()

    def set(self, name, arr):
        a = _as(arr, np.float64)
        self._chk(lib().hprlp_solver_set_vector(self.h, name.encode(), a.ctypes.data_as(c_dbl_p), len(a)))

    def scalars(self):
        out = np.zeros(16)
        self._chk(lib().hprlp_solver_get_scalars(self.h, out.ctypes.data_as(c_dbl_p)))
        keys = ("b_scale", "c_scale", "norm_b", "norm_c", "norm_b_org", "norm_c_org", "sigma", "lambda_max",
                "setup_time", "scaling_time", "power_time", "power_iters", "kx", "ky")
        return dict(zip(keys, out))

    def info(self):
        out = (C.c_long * 8)()
        self._chk(lib().hprlp_solver_info(self.h, out))
        keys = ("m", "n", "nnz", "blocks_A", "blocks_AT", "grid_y", "grid_x", "tiled")
        d = dict(zip(keys, [int(v) for v in out]))
        d["reordered"] = bool(d["tiled"] & 8)  # set-up time locality ordering in place (csrc/reorder.cpp)
        d["tiled"] &= 7
        return d

    def describe(self):
        """Which kernel form runs on A and A^T (hprlp_solver_describe)."""
        buf = C.create_string_buffer(2048)
        L = lib()
        L.hprlp_solver_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self._chk(L.hprlp_solver_describe(self.h, buf, 2048))
        return buf.value.decode()

    def run(self, max_trace=4096):
        res = CResults()
        trace = (CTraceRow * max_trace)()
        nt = C.c_int(0)
        self._chk(lib().hprlp_solver_run(self.h, C.byref(res), trace, max_trace, C.byref(nt)))
        r = Results(res, getattr(self, "m_loc", self.model.m), getattr(self, "n_loc", self.model.n))
        r.trace = [{f: getattr(trace[i], f) for f, _ in CTraceRow._fields_} for i in range(nt.value)]
        return r

    def time_iterations(self, warmup, steps, mode=0):
        t = C.c_double(0); tx = C.c_double(0); ty = C.c_double(0)
        self._chk(lib().hprlp_solver_time_iterations(self.h, warmup, steps, mode, C.byref(t), C.byref(tx), C.byref(ty)))
        return dict(total_ms=t.value, xhalf_ms=tx.value, yhalf_ms=ty.value)
