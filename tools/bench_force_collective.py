#!/usr/bin/env python3
"""bench.py on ONE GPU with a 1-rank RCCL process group, so that the exchange step (the early asynchronous all-reduce
beside the encoder's backward, then the rest) really runs RCCL calls: DVAE_FORCE_COLLECTIVE=1 makes GradSumReducer
exchange in a 1-rank group.  What a concurrent RCCL kernel does to the level kernels is read from `encode_bwd_ms`.
usage: DVAE_FORCE_COLLECTIVE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python tools/bench_force_collective.py <bench args>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DVAE_FORCE_COLLECTIVE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench  # noqa: E402

bench.main()
dist.destroy_process_group()
