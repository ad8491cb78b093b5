import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "mujoco: pins the oracle to a real MuJoCo; the physics part skips where the `mujoco` wheel is absent")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle
