import os
import sys

import pytest

# the libraries read their ZMI_* tuning / test overrides only in a process that has ZMI_TUNING set (checked once, at the
# first call): the tests use them (segment sizes, queue limits, chunk sizes), a product process never calls getenv()
os.environ.setdefault("ZMI_TUNING", "1")
# launches of up to 512 streams take the multi-wave decode kernel in the product; the parity tests are mostly launches of a few
# dozen streams and must reach the one-wave-per-stream kernel -- the one the benchmark times -- so "more than 16 streams" selects
# it here, as it did until round 4 (tests of the multi-wave kernel on larger launches: test_multi_wave_decode_up_to_512_streams)
os.environ.setdefault("ZMI_INF_MW_MAX", "16")

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


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zlib_rs_amd.engine import Engine
    e = Engine()
    yield e
    e.close()
