import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _prebuild():
    """Everything the CPU suite builds in the tree on first use (the oracle's library and binaries, the host emulation of the
    device templates, the JNI harness, the product library itself), built ONCE, here, by the controller - the helpers build
    lazily and write their outputs in place, which two workers must not do at the same time.  Up to date: milliseconds."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "kafka_specification_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan", "tsan"])
    import host_emu
    host_emu.build()
    import importlib.util
    for name, fn in (("test_jni_shim", "build_harness"), ("test_sanitizers_cpu", "build_sanitized_emu")):
        # (loaded under another name: the test modules themselves stay for pytest to import, with its assertion rewriting)
        spec = importlib.util.spec_from_file_location("_kmc_prebuild_" + name, os.path.join(ROOT, "tests", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        getattr(mod, fn)()


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: ~730 tests, 10 - 16 min one after the other on this container's cores, most of it oracle
    searches, sanitizer builds and two-rank gloo runs that do not share anything) spreads over pytest-xdist workers when nobody
    said otherwise: 3 min.  The tests are written for that (free rendezvous ports, temporary directories); what they build in
    the tree is built before the workers start (_prebuild; if that fails the suite runs one after the other, as before).
    Only for exactly that marker expression - the `-m gpu` suite shares one device and stays serial - and only when -n was
    not given; KMC_TEST_JOBS=0 switches it off, =N picks the workers."""
    if (config.option.markexpr or "").strip() != "not gpu" or "PYTEST_XDIST_WORKER" in os.environ:
        return None
    if getattr(config.option, "numprocesses", 0) is not None or not config.pluginmanager.hasplugin("xdist"):
        return None   # -n given (or -p no:xdist): as asked
    if getattr(config.option, "usepdb", False) or getattr(config.option, "collectonly", False):
        return None
    jobs = os.environ.get("KMC_TEST_JOBS", "")
    n = int(jobs) if jobs.isdigit() else min(6, max(1, (os.cpu_count() or 1) - 2))
    if n > 1:
        try:
            _prebuild()
        except Exception as e:   # (the tests that need the missing piece will say so themselves)
            sys.stderr.write(f"tests/conftest.py: pre-build failed ({e}); running the suite one test after the other\n")
            return None
        config.option.numprocesses = n   # (xdist's own pytest_cmdline_main, which runs after this one, turns it into --dist load)
    return None
