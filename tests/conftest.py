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


# ---- layer-2 parity report: every test that compares against reference-generated fixtures records what it ACHIEVED
# (not just pass/fail); the table is printed in the terminal summary, also under -q, so that the GPU test record carries
# the fidelity numbers.
_PARITY_ROWS = []


def record_parity(what: str, stats: dict, threshold: float) -> None:
    _PARITY_ROWS.append((what, stats, threshold))


@pytest.fixture
def parity_report():
    return record_parity


def pytest_terminal_summary(terminalreporter):
    if not _PARITY_ROWS:
        return
    tr = terminalreporter
    tr.write_sep("-", "layer-2 parity vs reference fixtures (achieved / required)")
    tr.write_line(f"{'fixture':52s} {'within band':>12s} {'required':>9s} {'bit-identical':>14s} {'mean rel err':>13s} {'mean abs err':>13s}")
    for what, st, thr in _PARITY_ROWS:
        tr.write_line(f"{what:52s} {100 * st['within']:11.3f}% {100 * thr:8.2f}% {100 * st['bit_identical']:13.2f}% "
                      f"{st['mean_rel_err']:13.2e} {st['mean_abs_err']:13.2e}")
