"""mpconstellation_amd -- MI355X-native batched constellation MPC (hot path of
rgovindjee/mpconstellation: linearise/discretise + per-satellite finite-horizon solve, plus the
nonlinear rollouts either side of it), behind the reference's Python API."""
from .constants import Constants
from .satellite import Satellite
from .satellite_scale import SatelliteScale
from .linearize_discretize import Discretizer
from .optimizer import Optimizer, mpc_step_batch, solve_batch, solve_shared_tf, scp_iteration_batch, mpc_update_batch
from .control import (Controller, ConstantThrustController, ConstantTangentialThrustController,
                      SequenceController, OptimalController)
from .simulator import Simulator, propagate_batch
from .constellation_mpc import ConstellationMPC

__all__ = ["Constants", "Satellite", "SatelliteScale", "Discretizer", "Optimizer", "mpc_step_batch", "solve_batch", "solve_shared_tf", "scp_iteration_batch", "mpc_update_batch",
           "Controller", "ConstantThrustController", "ConstantTangentialThrustController", "SequenceController",
           "OptimalController", "Simulator", "propagate_batch", "ConstellationMPC"]
