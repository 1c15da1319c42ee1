import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# keep the oracle's OpenMP pool small and passive inside the test-suite
os.environ.setdefault("OMP_NUM_THREADS", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

P = (1 << 64) - (1 << 32) + 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def fe(pkg):
    return pkg.frontend


@pytest.fixture(scope="session")
def oracle():
    import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def ctx(pkg):
    """HIP context; GPU tests fail loudly (no skip, no fallback) when the library or device is missing."""
    return pkg.Context(0)


def rand_field(rng, shape):
    """Uniform-ish canonical Goldilocks elements, with edge values mixed in."""
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    flat = a.reshape(-1)
    edges = np.array([0, 1, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 63), 0xFFFFFFFF00000000], dtype=np.uint64)
    k = min(len(edges), flat.size)
    if flat.size:
        pos = rng.choice(flat.size, size=k, replace=False)
        flat[pos] = edges[:k]
    return a
