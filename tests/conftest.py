"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` runs here (no GPU): the oracle against its known answers and golden fixtures,
the host logic, the seeded inits, and that libnbody_hip.so loads and exports the whole C ABI.
`-m gpu` runs on a real MI355X: the parity tests proper, all through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def nb():
    import wgpu_n_body_amd as nb
    from wgpu_n_body_amd.build import build_native
    build_native()
    return nb


@pytest.fixture(scope="session")
def gpu(nb):
    """The package, on a box with a HIP device.  GPU tests fail (not skip) without one."""
    assert nb.device_count() > 0, "a -m gpu test ran on a machine without a HIP device"
    return nb
