import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _sync_torch_first(fn):
    """The engine launches on its own non-blocking HIP stream, which is not ordered against torch's stream: a test that
    fills a result buffer with torch (zeros / full / fill_, asynchronous kernels) and enqueues engine work right after
    would race with its own fill (seen once: a fast k = 1 query finished before the fill ran). A caller must order its
    streams; here every enqueue-type call first waits for torch's outstanding work."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        t = sys.modules.get("torch")
        if t is not None and t.cuda.is_available():
            t.cuda.synchronize()
        return fn(*a, **kw)
    return wrapper


@pytest.fixture(scope="session")
def pkg():
    import _pkg
    try:
        # torch brings its own HIP runtime; it must initialise BEFORE libtkspmv.so's (the image's ROCm) touches the device,
        # or torch later reports "No HIP GPUs are available" (seen when a test created an engine before its first .cuda())
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    mod = _pkg.load()
    for name in ("enqueue", "enqueue_many", "enqueue_batch", "enqueue_multi", "time_queries", "time_multi"):
        fn = getattr(mod.SpMV, name, None)
        if fn is not None and not getattr(fn, "_torch_synced", False):
            w = _sync_torch_first(fn)
            w._torch_synced = True
            setattr(mod.SpMV, name, w)
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.oracle()
    return oracle_lib
