import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
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


@pytest.fixture(scope="session")
def meta():
    with open(os.path.join(GOLDEN, "problem_meta.json")) as fh:
        return json.load(fh)


def make_cfg(N=20, **kw):
    from trajtrack_mpcndqn_rlboost_amd import MpcConfig
    return MpcConfig(N_hor=N, **kw)


def oracle_cfg(cfg):
    import oracle
    return oracle.OracleConfig.from_dict(cfg.solver_dict())


@pytest.fixture(scope="session")
def cfg20():
    return make_cfg(20)


@pytest.fixture(scope="session")
def cfg40():
    return make_cfg(40)


@pytest.fixture(scope="session")
def solver20(cfg20):
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver
    s = BatchSolver(cfg20)
    yield s
    s.close()


@pytest.fixture(scope="session")
def solver40(cfg40):
    from trajtrack_mpcndqn_rlboost_amd import BatchSolver
    s = BatchSolver(cfg40)
    yield s
    s.close()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))
