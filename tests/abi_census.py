"""Test infrastructure: count the calls that reach each export of librcw_hip through the ctypes binding.

`install()` wraps every declared entry point on the loaded library objects (the shipped build and, where present, the
development build) with a counting closure; the binding hands out those same objects to every user in the process.  A
process that sets RCW_ABI_CALL_LOG (the GPU suite's conftest does, for itself and for the rank scripts it starts)
appends its counts to that file at exit; tests/test_zz_abi_call_coverage.py adds them up."""
import atexit
import collections
import json
import os

CALLS = collections.Counter()
_wrapped = set()


def _wrap(lib, names):
    if id(lib) in _wrapped:
        return
    _wrapped.add(id(lib))
    for name in names:
        fn = getattr(lib, name)

        def counted(*args, _fn=fn, _name=name):
            CALLS[_name] += 1
            return _fn(*args)

        setattr(lib, name, counted)


def install():
    from raycastworlds_jl_amd import _capi

    _wrap(_capi.load(), _capi.SIGNATURES)
    if os.path.exists(_capi.DEV_LIB_PATH):
        _wrap(_capi.load("dev"), _capi.SIGNATURES)
    log = os.environ.get("RCW_ABI_CALL_LOG")
    if log:
        atexit.register(dump, log)


def dump(path):
    if CALLS:
        with open(path, "a") as f:
            f.write(json.dumps(dict(CALLS)) + "\n")


def install_if_asked():
    """For the rank scripts the suite starts: count only when the suite asked for it."""
    if os.environ.get("RCW_ABI_CALL_LOG"):
        install()
