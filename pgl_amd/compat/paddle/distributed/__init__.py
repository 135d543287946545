"""paddle.distributed.{get_world_size, get_rank, init_parallel_env}: one process per GPU over torch.distributed (RCCL)."""
import os

import torch.distributed as _dist


def get_world_size():
    return _dist.get_world_size() if (_dist.is_available() and _dist.is_initialized()) else int(os.environ.get("WORLD_SIZE", "1"))


def get_rank():
    return _dist.get_rank() if (_dist.is_available() and _dist.is_initialized()) else int(os.environ.get("RANK", "0"))


def init_parallel_env():
    from pgl_amd.distributed import init_parallel_env as _init
    _init()


class ParallelEnv(object):
    @property
    def dev_id(self):
        return int(os.environ.get("LOCAL_RANK", "0"))

    local_rank = property(lambda self: get_rank())
    nranks = property(lambda self: get_world_size())
