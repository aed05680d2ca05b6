import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_ABI_CALLS = {}


def pytest_sessionstart(session):
    """KERNEL_COVERAGE_OUT=<file>: also count the calls of every C-ABI entry point made through the Python binding."""
    if not os.environ.get("KERNEL_COVERAGE_OUT"):
        return
    from gym_art_amd import _lib
    lib = _lib.load()
    for name, _res, _args in _lib.SYMBOLS:
        fn = getattr(lib, name)
        _ABI_CALLS[name] = 0

        def counted(*a, _fn=fn, _name=name):
            _ABI_CALLS[_name] += 1
            return _fn(*a)
        setattr(lib, name, counted)


def pytest_sessionfinish(session, exitstatus):
    """KERNEL_COVERAGE_OUT=<file>: which step / rollout kernel instantiations this test process launched (tools/kernel_coverage.py)."""
    out = os.environ.get("KERNEL_COVERAGE_OUT")
    if not out or "gym_art_amd._lib" not in sys.modules:
        return
    import ctypes as C
    import json
    lib = sys.modules["gym_art_amd._lib"].load()
    rec = {}
    for kind, name in ((0, "step"), (1, "rollout")):
        buf = (C.c_uint32 * 1024)()
        k = lib.gaq_launched_variants(kind, buf, 1024)
        rec[name] = [int(buf[i]) for i in range(min(k, 1024))]
    prev = {}
    if os.path.exists(out):                      # several pytest processes may add to one report
        prev = json.load(open(out))
    for name in rec:
        rec[name] = sorted(set(rec[name]) | set(prev.get(name, [])))
    rec["abi_calls"] = {k: v + prev.get("abi_calls", {}).get(k, 0) for k, v in _ABI_CALLS.items()}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(rec, open(out, "w"))
