import os
import sys

# numpy / scipy ship an OpenBLAS built for at most 64 threads; on hosts with more cores (the GPU boxes) its thread pool has crashed the test process
# (segmentation faults inside scipy.linalg.lu_factor -- the threaded getrf -- about once in fifteen runs of the GPU suite, also with 8 threads, and
# once long after a BLAS call).  ONE thread from the moment the library loads: the parallel LU path is never entered --
# this only takes effect if numpy has not been imported yet, the session fixture below covers the other case.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
# ... and still about once in fifteen runs with one thread, always inside dgetrf: the GPU hosts are Zen 5 (EPYC 9575F), for which these OpenBLAS
# builds (0.3.28 / 0.3.29, DYNAMIC_ARCH) select their AVX-512 "SkylakeX" kernels.  The AVX2 "Haswell" kernels are the ones every other x86 host runs.
os.environ.setdefault("OPENBLAS_CORETYPE", "Haswell")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session", autouse=True)
def _limit_blas_threads():
    """numpy / scipy ship an OpenBLAS built for at most 64 threads; on hosts with more cores (the GPU boxes) its threaded LU has crashed the test
    process (segmentation fault inside scipy.linalg.lu_factor, tests/ref_ipm.py).  The test infrastructure runs its BLAS single-threaded (a 780 x 780 LU takes 30 ms)."""
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            yield
    except ImportError:
        yield
