"""MI355X-native batched AC power-flow env.step() -- drop-in for the hot path of
danieleschmidt/grid-fed-rl-gym (GridEnvironment.step / NewtonRaphsonSolver.solve).

Host side is plain Python + NumPy over a C-ABI shared library (libgridstep.so, ctypes);
all arithmetic on the path runs in hand-written HIP kernels for gfx950.  There is no CPU
fallback: importing the solver/environment classes works anywhere, but constructing them
without the built library or without a GPU raises.
"""
from .components import (Bus, Line, Load, PowerFlowSolution, BatchedPowerFlowSolution,
                         PowerFlowError, InvalidActionError)
from .feeders import (FeederSpec, flatten_feeder, flatten_network, to_objects, reference_env_network,
                      with_reference_env_renewables, simple_radial, ieee13_like, ieee123_like,
                      random_meshed, scalable_like)

from .solver import (BatchedNewtonRaphsonSolver, NewtonRaphsonSolver, FastDecoupledSolver,
                     BatchedForwardBackwardSweepSolver, BatchedRobustPowerFlowSolver, DistributionPowerFlow, parallel_power_flow_batch,
                     injections_from_dicts)
from .env import BatchedGridEnvironment, VectorizedEnvironment, Box
from .rollout import collect_random_data, GridDataset
from .sharding import LoopbackShards, ShardedGridEnvironment, shard_range
from .multi_agent import AgentConfig, BatchedMultiAgentWrapper
from .feeders import feeder_from_dict, feeder_to_dict, network_dict_normalized
from .safety import BatchedSafetyChecker, BatchedSafetyMonitor, PostStepChecks, device_quality_score
from .unbalanced import UnbalancedPowerFlow, UnbalancedFeederSpec, UnbalancedSolution, unbalanced_from_single_phase, ieee8500_like

__all__ = [
    "BatchedNewtonRaphsonSolver", "NewtonRaphsonSolver", "FastDecoupledSolver",
    "BatchedForwardBackwardSweepSolver", "BatchedRobustPowerFlowSolver", "DistributionPowerFlow", "parallel_power_flow_batch",
    "injections_from_dicts", "BatchedGridEnvironment", "VectorizedEnvironment", "Box",
    "collect_random_data", "GridDataset", "ShardedGridEnvironment", "LoopbackShards", "shard_range",
    "AgentConfig", "BatchedMultiAgentWrapper", "feeder_from_dict", "feeder_to_dict", "network_dict_normalized",
    "BatchedSafetyChecker", "BatchedSafetyMonitor", "PostStepChecks", "device_quality_score",
    "UnbalancedPowerFlow", "UnbalancedFeederSpec", "UnbalancedSolution", "unbalanced_from_single_phase", "ieee8500_like",
    "Bus", "Line", "Load", "PowerFlowSolution", "BatchedPowerFlowSolution", "PowerFlowError",
    "InvalidActionError", "FeederSpec", "flatten_feeder", "flatten_network", "to_objects",
    "reference_env_network", "with_reference_env_renewables", "simple_radial", "ieee13_like",
    "ieee123_like", "random_meshed", "scalable_like",
]
__version__ = "0.1.0"
