"""dvae_amd -- MI355X-native D-VAE train-step hot path.

The directory is called `d-vae_amd` (not an importable identifier); load it with
`from _dvae_pkg import dvae_amd` (repo root helper) which registers it in
sys.modules as `dvae_amd`.
"""
from .batching import DAGBatch, GraphArrays, graphs_to_arrays, shard_bounds, balanced_shard_bounds  # noqa: F401
from .models import DVAE, DVAE_BN, PARAM_NAMES  # noqa: F401
from . import _lib, data, dist, graph, harness, optim  # noqa: F401
from .graph import DAG  # noqa: F401

__all__ = ["DVAE", "DVAE_BN", "DAG", "DAGBatch", "GraphArrays", "graphs_to_arrays", "shard_bounds", "balanced_shard_bounds", "PARAM_NAMES"]
