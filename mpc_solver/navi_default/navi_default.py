"""Drop-in for the OpEn-generated PyO3 module `navi_default` (optimizer_name of config/mpc_default.yaml).

The reference loads it with
    sys.path.append(os.path.join(root_dir, config.build_directory, config.optimizer_name))
    built_solver = __import__(config.optimizer_name); solver = built_solver.solver()
(src/mpc_traj_tracker/trajectory_generator.py:63-71) and calls `solver.run(p, initial_guess)` (:318).
Copy this directory to `<cwd>/mpc_solver/navi_default/` of the reference scripts (they run from `src/`), with the
repository root of this build on PYTHONPATH.  The yaml that defines the problem sizes is taken from the
environment variable MPCGPU_CONFIG, else from this build's config/mpc_default.yaml.
"""
import os

from trajtrack_mpcndqn_rlboost_amd import MpcConfig, Solver, default_config_path


def solver(device: int = 0) -> Solver:
    cfg = MpcConfig(os.environ.get("MPCGPU_CONFIG", default_config_path("mpc_default.yaml")))
    return Solver(cfg, device=device)
