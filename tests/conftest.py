import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: a GPU test that runs for tens of seconds (long-run drift checks)")


def _has_gpu():
    # counting devices does not initialise the GPU on this image
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def refcpu_mod():
    from oracle import refcpu
    refcpu.build()
    return refcpu


@pytest.fixture(scope="session")
def reflib_mod():
    from oracle import reflib
    if not reflib.available():
        pytest.skip("oracle/_ref/libmaniac_ref.so not built (needs /root/reference + amdflang)")
    return reflib


@pytest.fixture(autouse=True)
def _close_leftover_farm():
    """A test that fails before farm.close() must not leave the process-wide Fortran farm occupied."""
    yield
    mod = sys.modules.get("maniac_mc_amd.fortran_host")
    if mod is not None:
        for farm in list(mod.FortranFarm._slots.values()):
            farm.close()
