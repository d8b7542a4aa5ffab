"""pytest configuration: `-m gpu` = parity tests that need an MI355X; everything else runs on CPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as graft
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    import __graft_entry__ as graft
    return graft.load_oracle().Oracle()


@pytest.fixture(scope="session")
def oracle_truediv():
    import __graft_entry__ as graft
    return graft.load_oracle().Oracle(true_division=True)


@pytest.fixture(scope="session")
def native_lib(pkg):
    """libmi355pt.so, built in-tree by __graft_entry__.build() (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as graft
    if not os.path.exists(pkg.native.LIB_PATH):
        graft.build()
    return pkg.native.load()
