"""Configuration surface of the tracker: the same flat YAML keys as the reference.

Reference: ``config/mpc_default.yaml:7-55`` read by ``src/util/mpc_config.py:8-19`` (every YAML key becomes
an attribute).  The solver hyper-parameters that the reference hard-codes in
``src/mpc_traj_tracker/mpc/mpc_generator.py:22,285-293`` (and the opengen defaults it quotes in comments) are
accepted as OPTIONAL extra keys so that existing YAML files keep working unchanged.
"""
from __future__ import annotations

import os
from typing import Any, Dict

import yaml

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Optional solver keys and their defaults (mpc_generator.py:285-293).
SOLVER_DEFAULTS: Dict[str, Any] = dict(
    solver_tolerance=1e-4,               # tolerance
    solver_delta_tolerance=1e-4,         # constraints tolerance
    solver_initial_tolerance=1e-4,       # initial inner tolerance
    solver_initial_penalty=10.0,         # with_initial_penalty(10), mpc_generator.py:286
    solver_penalty_update_factor=5.0,
    solver_tolerance_update_factor=0.1,
    solver_sufficient_decrease=0.1,
    solver_max_inner_iterations=500,
    solver_max_outer_iterations=10,
    solver_lbfgs_memory=10,
    solver_max_duration_micros=5_000_000,  # MAX_SOVLER_TIME, mpc_generator.py:22
    fleet_weight=1000.0,                 # weight=1000, mpc_generator.py:216
    # What follows 10 line-search halvings without acceptance [OpEn; cannot be checked against an OpEn build here]:
    # 'last_trial' = the tau = 2^-10 trial point becomes the iterate (effective behaviour of the published engine),
    # 'half_step'  = tau = 0, the point u - gamma*fpr is evaluated and taken.  DESIGN.md section 3.
    solver_linesearch_fallback="last_trial",
    # When the ALM / PM loop keeps the penalty instead of multiplying it by solver_penalty_update_factor [OpEn; same caveat]:
    # 'either' = first outer iteration, or ||y+ - y|| OR ||F2|| shrank by solver_sufficient_decrease (the published engine's
    #            is_penalty_stall_criterion as recalled), 'both' = only when both shrank.  DESIGN.md section 3.
    solver_penalty_stall="either",
)

REQUIRED_KEYS = ("ts", "N_hor", "nu", "ns", "nq", "Nother", "Nstcobs", "nstcobs", "Ndynobs", "ndynobs",
                 "vehicle_width", "social_margin", "lin_vel_min", "lin_vel_max", "lin_acc_min", "lin_acc_max",
                 "ang_vel_max", "ang_acc_max")


def default_config_path(name: str = "mpc_default.yaml") -> str:
    return os.path.join(_REPO_ROOT, "config", name)


class MpcConfig:
    """Attribute bag with the reference's YAML keys (``Configurator`` semantics) plus solver defaults."""

    def __init__(self, yaml_fp: str | None = None, verbose: bool = False, **overrides):
        self._prtname = "[MPC-CFG]"
        yaml_fp = yaml_fp or default_config_path()
        if verbose:
            print(f'{self._prtname} Loading configuration from "{yaml_fp}".')
        with open(yaml_fp, "r") as stream:
            loaded = yaml.safe_load(stream)
        for key, val in SOLVER_DEFAULTS.items():
            setattr(self, key, val)
        for key, val in loaded.items():
            setattr(self, key, val)
        for key, val in overrides.items():
            setattr(self, key, val)
        missing = [k for k in REQUIRED_KEYS if not hasattr(self, k)]
        if missing:
            raise KeyError(f"{yaml_fp}: missing configuration keys {missing}")
        if verbose:
            print(f"{self._prtname} Configuration done.")

    # -- derived -------------------------------------------------------------------------------
    @property
    def num_params(self) -> int:
        """len(p): mpc_generator.py:179-188."""
        N = self.N_hor
        return (2 * self.ns + self.nu) + self.nq + (self.ns * N + N) + self.ns * N * self.Nother + \
            self.Nstcobs * self.nstcobs + self.Ndynobs * self.ndynobs * N + 2 * N

    @property
    def num_decision(self) -> int:
        return self.nu * self.N_hor

    def offsets(self) -> Dict[str, int]:
        """Start index of every block of the parameter vector."""
        N = self.N_hor
        r0 = 2 * self.ns + self.nu + self.nq
        c0 = r0 + self.ns * N + N
        os0 = c0 + self.ns * N * self.Nother
        od0 = os0 + self.Nstcobs * self.nstcobs
        qs0 = od0 + self.Ndynobs * self.ndynobs * N
        qd0 = qs0 + N
        return dict(s=0, q=2 * self.ns + self.nu, r=r0, vref=r0 + self.ns * N, c=c0, os=os0, od=od0, qstc=qs0,
                    qdyn=qd0, end=qd0 + N)

    def solver_dict(self, device: int = 0) -> Dict[str, Any]:
        """Fields of ``mpcgpu_config`` (include/mpcgpu.h) / of the oracle's config struct."""
        return dict(
            N=int(self.N_hor), nu=int(self.nu), ns=int(self.ns), Nother=int(self.Nother),
            Nstcobs=int(self.Nstcobs), nstcobs=int(self.nstcobs), Ndynobs=int(self.Ndynobs),
            ndynobs=int(self.ndynobs), ts=float(self.ts),
            lin_vel_min=float(self.lin_vel_min), lin_vel_max=float(self.lin_vel_max),
            ang_vel_max=float(self.ang_vel_max), lin_acc_min=float(self.lin_acc_min),
            lin_acc_max=float(self.lin_acc_max), ang_acc_max=float(self.ang_acc_max),
            vehicle_width=float(self.vehicle_width), social_margin=float(self.social_margin),
            fleet_weight=float(self.fleet_weight),
            tol=float(self.solver_tolerance), delta_tol=float(self.solver_delta_tolerance),
            init_tol=float(self.solver_initial_tolerance), init_penalty=float(self.solver_initial_penalty),
            penalty_update=float(self.solver_penalty_update_factor),
            tol_update=float(self.solver_tolerance_update_factor),
            suff_decrease=float(self.solver_sufficient_decrease),
            max_inner=int(self.solver_max_inner_iterations), max_outer=int(self.solver_max_outer_iterations),
            lbfgs_mem=int(self.solver_lbfgs_memory), device=int(device),
            max_duration_us=float(self.solver_max_duration_micros),
            ls_fallback=1 if self.solver_linesearch_fallback == "half_step" else 0,
            stall_rule=1 if self.solver_penalty_stall == "both" else 0)


# The reference's class name (src/util/mpc_config.py:8): same constructor, same attribute semantics.
Configurator = MpcConfig
