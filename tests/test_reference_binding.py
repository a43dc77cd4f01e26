"""N4: the reference's OWN pybind11 module (compiled by `make -C oracle refbinding` from
/root/reference/bindings/python/src/hprlp_pybind.cpp, unchanged) running on top of lib/libhprlp.so.
Shows that the existing Python binding works unchanged against the MI355X library."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

CAND = glob.glob(os.path.join(ROOT, "oracle", "_ref", "_hprlp_core*.so"))


def core():
    if not CAND:
        pytest.skip("oracle/_ref/_hprlp_core*.so not built (needs /root/reference: make -C oracle refbinding)")
    spec = importlib.util.spec_from_file_location("_hprlp_core", CAND[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_model(c):
    inf = np.inf
    return c.create_model_from_arrays(2, 2, 4, np.array([0, 2, 4], np.int32), np.array([0, 1, 0, 1], np.int32),
                                      np.array([1.0, 2.0, 3.0, 1.0]), np.array([-inf, -inf]), np.array([10.0, 12.0]),
                                      np.array([0.0, 0.0]), np.array([inf, inf]), np.array([-3.0, -5.0]), False)


def test_reference_pybind_builds_models_on_our_library():
    c = core()
    model = make_model(c)
    assert model.is_valid() and (model.m, model.n) == (2, 2)
    p = c.Parameters()
    assert (p.max_iter, p.stop_tol, p.check_iter) == (2**31 - 1, 1e-4, 150)
    m2 = c.create_model_from_mps(os.path.join(ROOT, "tests", "data", "lp_small.mps"))
    assert (m2.m, m2.n) == (2, 2)
    c.free_model(model)
    c.free_model(m2)


@pytest.mark.gpu
def test_reference_pybind_solves_on_gpu(gpu):
    c = core()
    model = make_model(c)
    p = c.Parameters()
    p.stop_tol = 1e-9
    p.use_presolve = False
    r = c.solve(model, p)
    assert r.status == "OPTIMAL" and abs(r.primal_obj + 26.4) < 1e-6
    np.testing.assert_allclose(np.asarray(r.x), [2.8, 3.6], atol=1e-6)
    c.free_model(model)
