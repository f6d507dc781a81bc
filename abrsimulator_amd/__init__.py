"""MI355X-native batched ABR environment + MPC lookahead.

Importing the package loads the HIP library eagerly; if it has not been built
the import fails (there is no CPU fallback).
"""
from . import _lib
from .datamodel import MPD, Chunk, ChunkInfo, NetworkInfo, QOEMetric
from .env import BatchedABREnv, obs_dict, pack_traces
from .mpc import BatchedMPCController, EnvPlayer
from .sharding import ShardedABREnv, ShardStep
from .simulator import Simulator
from .traces import (load_mpd_file, load_network_info, load_trace_file, save_mpd_file,
                     save_trace_file)

_lib.lib()   # fail loudly at import time when libabr_hip.so is missing

__all__ = ["MPD", "Chunk", "ChunkInfo", "NetworkInfo", "QOEMetric", "BatchedABREnv",
           "BatchedMPCController", "EnvPlayer", "obs_dict", "pack_traces", "Simulator", "ShardedABREnv", "ShardStep",
           "load_trace_file", "load_network_info", "load_mpd_file", "save_trace_file",
           "save_mpd_file"]
