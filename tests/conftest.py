import os
import sys

import pytest

# libzang_hip.so remembers its form switches (ZH_*_RANGES, ZH_*_PC_MAX, ...) after the first look-up unless this is set when
# it is loaded: the parity tests flip switches between paints to force every kernel form (csrc/ctx.hip zh_env)
os.environ["ZH_ENV_LIVE"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import zang_amd
    return zang_amd.default_context()
