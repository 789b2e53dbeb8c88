"""MI355X-native batched NMPC trajectory tracker (hot path of Woodenonez/TrajTrack-MPCnDQN-RLBoost).

The product is ``libmpcgpu.so`` (hand-written HIP for gfx950 behind the C-ABI of ``include/mpcgpu.h``);
this package is the thin Python host side that mirrors the reference's solver-plugin / tracker interface:

* :class:`MpcConfig`           -- the ``mpc_*.yaml`` config surface (reference ``util/mpc_config.py:8-19``)
* :class:`BatchSolver`         -- ctypes binding of the C-ABI (batched ``solver.run``)
* :func:`solver` / ``Solver``  -- the OpEn plugin contract ``<optimizer_name>.solver().run(p, initial_guess)``
* :class:`InterfaceMpc` / :class:`TrajectoryGenerator` -- the reference's tracker harness API (single robot)
* :class:`BatchedTracker`      -- the same harness for B robots with one GPU solve per tick
* :class:`DeviceTracker`       -- ... with device-resident state: assembly, solve and rollouts enqueued, nothing read back
* :mod:`scenes`                -- seeded synthetic parameter vectors for benchmarks and tests

Nothing here falls back to a CPU implementation: without the built library and a HIP device every solve
raises.
"""
from .config import MpcConfig, Configurator, default_config_path
from .solver import BatchSolver, BatchResult, MpcGpuError, STATUS_NAMES, library_path, build_library
from .plugin import Solver, SolverStatus, solver
from .interface_mpc import InterfaceMpc, TrajectoryTracker
from .trajectory_generator import TrajectoryGenerator
from .batched_tracker import BatchedTracker
from .device_tracker import DeviceTracker
from .motion_model import unicycle_model

__all__ = ["MpcConfig", "Configurator", "default_config_path", "BatchSolver", "BatchResult", "MpcGpuError",
           "STATUS_NAMES", "library_path", "build_library", "Solver", "SolverStatus", "solver", "InterfaceMpc",
           "TrajectoryTracker", "TrajectoryGenerator", "BatchedTracker", "DeviceTracker", "unicycle_model"]
