import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    # The whole suite under the OTHER reading of the ALM penalty-stall rule: every MpcConfig the tests build
    # -- and with it the GPU handle AND the oracle config derived from it -- takes this value unless a test names one itself.
    # `pytest --penalty-stall both`, or MPC_TEST_PENALTY_STALL=both in the environment.
    parser.addoption("--penalty-stall", choices=("either", "both"), default=None,
                     help="default of the yaml key solver_penalty_stall for this run")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    stall = config.getoption("--penalty-stall") or os.environ.get("MPC_TEST_PENALTY_STALL")
    if stall:
        assert stall in ("either", "both"), stall
        from trajtrack_mpcndqn_rlboost_amd import config as pkg_config
        pkg_config.SOLVER_DEFAULTS["solver_penalty_stall"] = stall


def pytest_report_header(config):
    from trajtrack_mpcndqn_rlboost_amd import config as pkg_config
    return f"solver_penalty_stall default of this run: {pkg_config.SOLVER_DEFAULTS['solver_penalty_stall']}"


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


# Tests about answers that REST ON a hard constraint (||F2|| driven below delta by a growing penalty: the "grazing" family, a lone
# disc on the path) need the penalty to grow while the acceleration constraints are inactive -- which only the "both" reading of
# the penalty-stall rule does (under "either" y+ = y = 0 keeps c at its initial 10, the solve ends at the outer
# cap with ||F2|| ~ 1e-2 and nothing "converges on the constraint"; test_solution_kkt.py asserts exactly that).  They name it.
GROWING_PENALTY = dict(solver_penalty_stall="both")


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
