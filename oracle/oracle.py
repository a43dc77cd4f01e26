"""ctypes loader for the CPU oracle (oracle/hpr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing in the product (hpr-lp-c_amd/, lib/libhprlp.so) imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborc_hprlp.so")

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class Params(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int),
        ("stop_tol", C.c_double),
        ("time_limit", C.c_double),
        ("check_iter", C.c_int),
        ("use_CR_scaling", C.c_int),
        ("use_Ruiz_scaling", C.c_int),
        ("use_Pock_Chambolle_scaling", C.c_int),
        ("use_bc_scaling", C.c_int),
    ]

    @classmethod
    def default(cls, **kw):
        # defaults of reference include/structs.h:26-39
        p = cls(2**31 - 1, 1e-4, 3600.0, 150, 1, 1, 1, 1)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class ScalingScalars(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("b_scale", "c_scale", "norm_b", "norm_c", "norm_b_org", "norm_c_org")]


class TraceRow(C.Structure):
    _fields_ = [("iter", C.c_int), ("restart_flag", C.c_int)] + [
        (k, C.c_double)
        for k in ("err_Rp", "err_Rd", "primal_obj", "dual_obj", "gap", "kkt", "sigma", "current_gap", "lambda_max",
                  "last_gap", "save_gap", "inner")
    ]


class Result(C.Structure):
    _fields_ = [
        ("residuals", C.c_double), ("primal_obj", C.c_double), ("gap", C.c_double),
        ("time4", C.c_double), ("time6", C.c_double), ("time8", C.c_double), ("time", C.c_double),
        ("iter4", C.c_int), ("iter6", C.c_int), ("iter8", C.c_int), ("iter", C.c_int),
        ("status", C.c_char * 64),
        ("lambda_max", C.c_double), ("power_time", C.c_double), ("power_iters", C.c_int),
        ("n_trace", C.c_int), ("n_restarts", C.c_int),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("hpr_oracle.c", "hpr_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_power_iteration.restype = C.c_double
        _lib.orc_time_iterations.restype = C.c_double
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_int_p)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dbl_p)


def transpose(m, n, rp, ci, v):
    nnz = len(v)
    rp, prp = _i(rp); ci, pci = _i(ci); v, pv = _d(v)
    trp = np.zeros(n + 1, np.int32); tci = np.zeros(max(nnz, 1), np.int32); tv = np.zeros(max(nnz, 1))
    lib().orc_csr_transpose(m, n, nnz, prp, pci, pv, trp.ctypes.data_as(c_int_p), tci.ctypes.data_as(c_int_p),
                            tv.ctypes.data_as(c_dbl_p))
    return trp, tci[:nnz], tv[:nnz]


def spmv(rows, rp, ci, v, x):
    rp, prp = _i(rp); ci, pci = _i(ci); v, pv = _d(v); x, px = _d(x)
    y = np.zeros(rows)
    lib().orc_spmv(rows, prp, pci, pv, px, y.ctypes.data_as(c_dbl_p))
    return y


def power_start_vector(m, seed=1, offset=0):
    z = np.zeros(m)
    lib().orc_power_start_vector(m, C.c_ulonglong(seed), C.c_longlong(offset), z.ctypes.data_as(c_dbl_p))
    return z


class ScaledLP:
    """Scaled copy of an LP (what the reference holds on the device after scaling())."""

    def __init__(self, m, n, rp, ci, v, AL, AU, l, u, c, params=None):
        params = params or Params.default()
        self.m, self.n = m, n
        self.Arp = np.ascontiguousarray(rp, np.int32)
        self.Aci = np.ascontiguousarray(ci, np.int32)
        self.Av = np.array(v, np.float64)
        self.ATrp, self.ATci, self.ATv = transpose(m, n, self.Arp, self.Aci, self.Av)
        self.ATv = np.array(self.ATv)
        self.AL = np.array(AL, np.float64); self.AU = np.array(AU, np.float64)
        self.l = np.array(l, np.float64); self.u = np.array(u, np.float64); self.c = np.array(c, np.float64)
        self.row_norm = np.zeros(m); self.col_norm = np.zeros(n)
        self.sc = ScalingScalars()
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        I = lambda a: a.ctypes.data_as(c_int_p)
        lib().orc_scaling(m, n, I(self.Arp), I(self.Aci), P(self.Av), I(self.ATrp), I(self.ATci), P(self.ATv),
                          P(self.AL), P(self.AU), P(self.l), P(self.u), P(self.c), C.byref(params),
                          P(self.row_norm), P(self.col_norm), C.byref(self.sc))

    def power_iteration(self, z0=None, max_iter=5000, tol=1e-4):
        if z0 is None:
            z0 = power_start_vector(self.m)
        z0 = np.ascontiguousarray(z0, np.float64)
        it = C.c_int(0)
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        I = lambda a: a.ctypes.data_as(c_int_p)
        lam = lib().orc_power_iteration(self.m, self.n, I(self.Arp), I(self.Aci), P(self.Av), I(self.ATrp),
                                        I(self.ATci), P(self.ATv), P(z0), max_iter, C.c_double(tol), C.byref(it))
        return lam, it.value

    def x_half(self, st, sigma, k, check):
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        I = lambda a: a.ctypes.data_as(c_int_p)
        lib().orc_x_half(self.n, I(self.ATrp), I(self.ATci), P(self.ATv), P(st["y"]), P(st["x"]), P(st["x_hat"]),
                         P(st["x_bar"]), P(st["z_bar"]), P(st["x_temp"]), P(self.l), P(self.u), P(self.c),
                         P(st["last_x"]), C.c_double(sigma), int(k), int(check))

    def y_half(self, st, sigma, lambda_max, k, check):
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        I = lambda a: a.ctypes.data_as(c_int_p)
        lib().orc_y_half(self.m, I(self.Arp), I(self.Aci), P(self.Av), P(st["x_hat"]), P(st["y"]), P(st["y_bar"]),
                         P(st["y_obj"]), P(st["y_temp"]), P(self.AL), P(self.AU), P(st["last_y"]),
                         C.c_double(sigma), C.c_double(lambda_max), int(k), int(check))

    def new_state(self):
        n, m = self.n, self.m
        st = {k: np.zeros(n) for k in ("x", "last_x", "x_hat", "x_bar", "z_bar", "x_temp")}
        st.update({k: np.zeros(m) for k in ("y", "last_y", "y_bar", "y_obj", "y_temp")})
        return st

    def time_iterations(self, sigma, lambda_max, iters):
        P = lambda a: a.ctypes.data_as(c_dbl_p)
        I = lambda a: a.ctypes.data_as(c_int_p)
        x = np.zeros(self.n); y = np.zeros(self.m)
        return lib().orc_time_iterations(self.m, self.n, I(self.Arp), I(self.Aci), P(self.Av), I(self.ATrp),
                                         I(self.ATci), P(self.ATv), P(self.AL), P(self.AU), P(self.l), P(self.u),
                                         P(self.c), C.c_double(sigma), C.c_double(lambda_max), int(iters), P(x), P(y))


def solve(m, n, rp, ci, v, AL, AU, l, u, c, obj_constant=0.0, params=None, lambda_override=0.0, max_trace=4096):
    params = params or Params.default()
    rp, prp = _i(rp); ci, pci = _i(ci); v, pv = _d(v)
    AL, pAL = _d(AL); AU, pAU = _d(AU); l, pl = _d(l); u, pu = _d(u); c, pc = _d(c)
    x = np.zeros(n); y = np.zeros(m); z = np.zeros(n)
    res = Result()
    trace = (TraceRow * max_trace)()
    lib().orc_solve(m, n, len(v), prp, pci, pv, pAL, pAU, pl, pu, pc, C.c_double(obj_constant), C.byref(params),
                    C.c_double(lambda_override), x.ctypes.data_as(c_dbl_p), y.ctypes.data_as(c_dbl_p),
                    z.ctypes.data_as(c_dbl_p), C.byref(res), trace, max_trace)
    rows = [
        {f: getattr(trace[i], f) for f, _ in TraceRow._fields_}
        for i in range(res.n_trace)
    ]
    out = {f: getattr(res, f) for f, _ in Result._fields_}
    out["status"] = res.status.decode()
    out.update(x=x, y=y, z=z, trace=rows)
    return out


def solve_batched(m, n, rp, ci, v, B, Cmat, AL, AU, L, U, obj_constants=None, model_obj_constant=0.0, params=None,
                  lambda_override=0.0):
    """Panels column-major (n x B / m x B) flattened, exactly as the C ABI takes them."""
    params = params or Params.default()
    rp, prp = _i(rp); ci, pci = _i(ci); v, pv = _d(v)
    Cmat, pC = _d(Cmat); AL, pAL = _d(AL); AU, pAU = _d(AU); L, pL = _d(L); U, pU = _d(U)
    if obj_constants is not None:
        obj_constants, pobjc = _d(obj_constants)
    else:
        pobjc = None
    X = np.zeros(n * B); Y = np.zeros(m * B); Z = np.zeros(n * B)
    pobj = np.zeros(B); resid = np.zeros(B); gap = np.zeros(B); it = np.zeros(B, np.int32)
    status = C.create_string_buffer(64 * B)
    lam = C.c_double(0)
    P = lambda a: a.ctypes.data_as(c_dbl_p)
    lib().orc_solve_batched(m, n, len(v), prp, pci, pv, B, pC, pAL, pAU, pL, pU, pobjc,
                            C.c_double(model_obj_constant), C.byref(params), C.c_double(lambda_override),
                            P(X), P(Y), P(Z), P(pobj), P(resid), P(gap), it.ctypes.data_as(c_int_p), status,
                            C.byref(lam))
    st = [status.raw[64 * k:64 * (k + 1)].split(b"\0")[0].decode() for k in range(B)]
    return dict(x=X.reshape(B, n), y=Y.reshape(B, m), z=Z.reshape(B, n), primal_obj=pobj, residuals=resid, gap=gap,
                iter=it, status=st, lambda_max=lam.value)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
