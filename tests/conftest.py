import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: ~730 tests, 10 - 16 min one after the other on this container's cores, most of it oracle
    searches, sanitizer builds and two-rank gloo runs that do not share anything) spreads over pytest-xdist workers when nobody
    said otherwise: 3 - 4 min.  The tests are written for that (free rendezvous ports, temporary directories, one builder per
    shared object under a file lock).  Only for exactly that marker expression - the `-m gpu` suite shares one device and stays
    serial - and only when -n was not given; KMC_TEST_JOBS=0 switches it off, =N picks the workers."""
    if (config.option.markexpr or "").strip() != "not gpu" or "PYTEST_XDIST_WORKER" in os.environ:
        return None
    if getattr(config.option, "numprocesses", 0) is not None or not config.pluginmanager.hasplugin("xdist"):
        return None   # -n given (or -p no:xdist): as asked
    if getattr(config.option, "usepdb", False) or getattr(config.option, "collectonly", False):
        return None
    jobs = os.environ.get("KMC_TEST_JOBS", "")
    n = int(jobs) if jobs.isdigit() else min(6, max(1, (os.cpu_count() or 1) - 2))
    if n > 1:
        config.option.numprocesses = n   # (xdist's own pytest_cmdline_main, which runs after this one, turns it into --dist load)
    return None
