"""Runs last (file name): every export of include/rcw.h was CALLED on the GPU by the suite — through the Python mirror in
this process, the rank scripts of the RCCL tests, or the plain-C harness (tests/c_abi_harness.c, whose calls are read off
its source)."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_every_export_was_called_on_the_gpu(request, _abi_call_census):
    import abi_census
    from raycastworlds_jl_amd import _capi

    if len(request.session.items) < 120 or request.config.option.keyword:
        pytest.skip("only meaningful after the whole GPU suite")
    total = dict(abi_census.CALLS)
    if _abi_call_census and os.path.exists(_abi_call_census):
        for line in open(_abi_call_census):
            for k, v in json.loads(line).items():
                total[k] = total.get(k, 0) + v
    harness = set(re.findall(r"\b(rcw_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "tests", "c_abi_harness.c")).read()))
    declared = set(re.findall(r"\b(rcw_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "rcw.h")).read()))
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    never = sorted(n for n in declared if total.get(n, 0) == 0 and n not in harness)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "abi_call_counts.json"), "w") as f:
        json.dump({"through_the_python_binding": dict(sorted(total.items())), "also_called_by_the_c_harness": sorted(harness & declared),
                   "never_called": never}, f, indent=1)
    assert not never, f"exports the GPU suite never called: {never}"
