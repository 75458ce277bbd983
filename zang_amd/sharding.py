"""Voice sharding across the GPUs of one node.

Voices share nothing (module state is per instance; Noise is seeded by GLOBAL voice index),
so painting needs no inter-GPU traffic.  The only exchange is the final mixdown: each rank
reduces its own voices to a [frames] partial on its GPU, then ONE sum all-reduce of that
4 KiB vector (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import torch.distributed as dist


def voice_range(total_voices, rank, world):
    """Contiguous shard [lo, hi) of `total_voices` for `rank` of `world`."""
    lo = total_voices * rank // world
    hi = total_voices * (rank + 1) // world
    return lo, hi


def allreduce_mix(mix, group=None):
    """Sum the per-rank partial mixes in place (float32 [frames], on the rank's device)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if mix.is_cuda and dist.get_backend(group) == "gloo":     # CPU-only backend (tests, dry runs): stage through host
            host = mix.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            mix.copy_(host)
        else:
            dist.all_reduce(mix, op=dist.ReduceOp.SUM, group=group)
    return mix
