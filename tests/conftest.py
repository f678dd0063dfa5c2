import os
import sys

import pytest

# the libraries read their ZMI_* tuning / test overrides only in a process that has ZMI_TUNING set (checked once, at the
# first call): the tests use them (segment sizes, queue limits, chunk sizes), a product process never calls getenv()
os.environ.setdefault("ZMI_TUNING", "1")
# Decode-kernel selection: the product gives every stream of a launch of up to 512 streams a 16-wave workgroup
# (zmi_inflate_kernel<*, 16>) and one wave per stream above that (the kernel the benchmark times).  The parity launches are a few
# dozen streams, so every test that decodes runs TWICE (fixture `inf_selection`): "product" = the library's own selection, nothing
# overridden; "onewave" = ZMI_INF_MW_MAX=16, which sends the same launches of 17+ streams through the one-wave kernel.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # no single test may hold the suite for long (a state-machine bug once looped for an hour): pytest-timeout, if there
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(params=["product", "onewave"])
def inf_selection(request, monkeypatch):
    """every inflate family under both decode kernels; the override is read per call (zmi_api.hip)"""
    if request.param == "onewave":
        monkeypatch.setenv("ZMI_INF_MW_MAX", "16")
    else:
        monkeypatch.delenv("ZMI_INF_MW_MAX", raising=False)
    return request.param


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zlib_rs_amd.engine import Engine
    e = Engine()
    yield e
    e.close()
