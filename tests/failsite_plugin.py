"""Test infrastructure (pytest plugin): which error returns of rcw_api.hip does a run provoke?

    RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so PYTHONPATH=tests python -m pytest tests -m gpu -q -p failsite_plugin

Every default load then takes the development build, whose fail() records its source line (rcw_dev_fail_sites, development build
only); at the end the explicit `fail(` sites never taken are listed in gpurun_out/failcov_<FAILCOV_TAG>.txt.  (The calls of the
rank scripts, the examples and the plain-C harness run in other processes and are not counted.)"""
import ctypes as C, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def pytest_sessionfinish(session, exitstatus):
    from raycastworlds_jl_amd import _capi
    lib = C.CDLL(_capi.DEV_LIB_PATH)            # (the same loaded object: dlopen returns the resident one)
    buf = (C.c_ubyte * 4096)()
    n = lib.rcw_dev_fail_sites(buf, 4096)
    hit = {i for i in range(n) if buf[i]}
    src = open(os.path.join(ROOT, "raycastworlds.jl_amd", "csrc", "rcw_api.hip")).read().splitlines()
    sites = [i + 1 for i, l in enumerate(src) if re.search(r"\bfail\(|RCW_HIP\(|RCW_TRY\(|RCW_NCCL\(", l) and not l.lstrip().startswith(("#define", "//", "int fail", "return fail(code"))]
    tag = os.environ.get("FAILCOV_TAG", "run")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"failcov_{tag}.txt"), "w") as f:
        expl = [s for s in sites if "fail(" in src[s - 1]]
        f.write(f"{len(hit)} lines recorded; explicit fail( sites: {len(expl)}, of them taken: {len([s for s in expl if s in hit])}\n")
        f.write("explicit sites never taken:\n")
        for s in expl:
            if s not in hit:
                f.write(f"  {s:5d}: {src[s - 1].strip()[:170]}\n")
        f.write("HIP / RCCL call sites whose failure branch was taken:\n")
        for s in sites:
            if s in hit and "fail(" not in src[s - 1]:
                f.write(f"  {s:5d}: {src[s - 1].strip()[:170]}\n")
