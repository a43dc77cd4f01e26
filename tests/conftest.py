"""pytest configuration: registers the `gpu` marker and loads the package by path
(the package directory `hpr-lp-c_amd/` is not a valid Python identifier)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The parity tests force kernel forms and thresholds through the library's TEST HOOKS (csrc/env.h): honoured only with this set.
# With no hook set the library runs its default path, as in production (tests/test_gpu_edge.py checks that a hook WITHOUT this
# is ignored and reported).
os.environ.setdefault("HPRLP_TEST_HOOKS", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _load(name, rel):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


hprlp = _load("hprlp_amd", os.path.join("hpr-lp-c_amd", "hprlp.py"))
lpgen = _load("hprlp_lpgen", os.path.join("hpr-lp-c_amd", "lpgen.py"))
shardlib = _load("hprlp_shard", os.path.join("hpr-lp-c_amd", "shard.py"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (never skip silently) when the HIP library is missing."""
    if not os.path.exists(hprlp.LIB_PATH):
        pytest.fail("lib/libhprlp.so missing on a GPU run: build with `make`")
    if not _have_gpu():
        pytest.fail("no /dev/kfd: -m gpu tests need the MI355X box")
    return hprlp.lib()


INF = float("inf")


@pytest.fixture(scope="session")
def model_mps_arrays():
    """The reference's only known-answer LP (reference data/model.mps, examples/c/example_direct_lp.c:20-33)."""
    return dict(m=2, n=2, rowptr=[0, 2, 4], colind=[0, 1, 0, 1], values=[1.0, 2.0, 3.0, 1.0],
                AL=[-INF, -INF], AU=[10.0, 12.0], l=[0.0, 0.0], u=[INF, INF], c=[-3.0, -5.0])


def _cpu_share():
    """CPUs this process may really use (cgroup quota, else affinity mask)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The oracle's OpenMP team defaults to every hardware thread of the host (256 on the GPU box, whose quota is 16
    CPUs): size it to the share, or the oracle-heavy tests mostly measure the scheduler."""
    try:
        from oracle import oracle as O
        O.set_num_threads(_cpu_share())
    except Exception:
        pass
    yield
