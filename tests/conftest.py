"""Shared fixtures.  `-m "not gpu"` = oracle vs golden vectors, host logic, C-ABI symbol
checks (no device calls).  `-m gpu` = parity of the HIP product against the oracle through
the C-ABI on a real MI355X."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_BIN = os.path.join(ORACLE_DIR, "bin", "pgx_oracle")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_bin():
    """The CPU restatement (test infrastructure). Built on demand with plain make.  PGX_ORACLE_SAN=1: the build under
    AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle san`), every report fatal."""
    if os.environ.get("PGX_ORACLE_SAN", "0") not in ("", "0"):
        san = os.path.join(ORACLE_DIR, "bin", "pgx_oracle_san")
        subprocess.check_call(["make", "-C", ORACLE_DIR, "san"], stdout=subprocess.DEVNULL)
        os.environ.setdefault("ASAN_OPTIONS", "abort_on_error=1:detect_leaks=0")
        os.environ.setdefault("UBSAN_OPTIONS", "halt_on_error=1:print_stacktrace=1")
        return san
    if not os.path.exists(ORACLE_BIN) or not os.path.exists(os.path.join(ORACLE_DIR, "liboracle.so")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so", "bin/pgx_oracle"],
                              stdout=subprocess.DEVNULL)
    return ORACLE_BIN


@pytest.fixture(scope="session")
def gold():
    return GOLD


def run_cmd(cmd, cwd=None, timeout=120):
    p = subprocess.run(cmd, cwd=cwd, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr
