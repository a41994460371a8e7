import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_launcher = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-rank GPU tests need fresh child processes, and a process that has initialised HIP must not exec:
    # start the (GPU-free) launcher now, before any test touches the device.
    global _launcher
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        _launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True)


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.close()
            _launcher.wait(timeout=10)
        except Exception:
            _launcher.kill()
        _launcher = None


@pytest.fixture(scope="session")
def launch_ranks():
    """launch_ranks(argv, n, env, timeout) -> (return codes, output tails) of n fresh rank processes."""
    def run(argv, n, env=None, timeout=600, raw=False):
        if _launcher is None:
            pytest.skip("rank launcher not started (run with -m gpu)")
        _launcher.stdin.write(json.dumps({"argv": argv, "n": n, "env": env or {}, "timeout": timeout, "raw": raw}) + "\n")
        _launcher.stdin.flush()
        rep = json.loads(_launcher.stdout.readline())
        return rep["rc"], rep["out"]
    return run


@pytest.fixture(scope="session")
def pkg():
    import gsr_pkg
    return gsr_pkg.load()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
