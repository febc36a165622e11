import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def rcw():
    """The product package (host mirror + ctypes binding of librcw_hip.so).  The shared library is a
    build artefact (git-ignored): compile it with hipcc if this checkout does not have it yet."""
    import raycastworlds_jl_amd as RCW
    from raycastworlds_jl_amd import _capi
    from raycastworlds_jl_amd import build as _build

    if not os.path.exists(_capi.LIB_PATH):
        _build.build()
    return RCW


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle — test infrastructure only."""
    from oracle import oracle as O

    O.build()
    # OpenMP would start a thread per hardware thread of the HOST (hundreds on a GPU box) inside a CPU share of 16: a step of
    # three agents then takes 100-200 ms in its barriers (measured, round 4) — most of the suite's time.  Sixteen at most.
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    O.set_num_threads(max(1, min(16, cpus)))
    return O


@pytest.fixture(scope="session", autouse=True)
def _abi_call_census(request, tmp_path_factory):
    """GPU runs: count the calls that reach each export of the library through the binding — in this process and, through
    RCW_ABI_CALL_LOG, in the rank scripts the suite starts (tests/abi_census.py; tests/test_zz_abi_call_coverage.py reads it)."""
    markexpr = request.config.option.markexpr or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        yield None
        return
    import abi_census

    log = str(tmp_path_factory.mktemp("abi") / "calls.jsonl")
    os.environ["RCW_ABI_CALL_LOG"] = log
    abi_census.install()
    yield log
    os.environ.pop("RCW_ABI_CALL_LOG", None)
