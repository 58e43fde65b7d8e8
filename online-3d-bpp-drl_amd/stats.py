"""Episode statistics kept on the device and reduced across GPUs.

Replaces the training loop's per-bin scan of `infos` (main.py:159-162) and is the ONLY thing that
ever crosses xGMI: bins are independent, so a multi-GPU job shards bins by global id and all-reduces
this 32-byte record (sum of episode returns, sum of final ratios, sum of episode lengths, episode
count) once per logging interval -- `torch.distributed` backend "nccl" (= RCCL) on GPUs, "gloo" in
CPU tests."""
import ctypes

import torch

from . import _lib


def shard_range(total_envs, rank, world_size):
    """Global bin ids [lo, hi) owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class EpisodeStats(object):
    FIELDS = ("return_sum", "ratio_sum", "length_sum", "episodes")

    def __init__(self, device):
        self.device = torch.device(device)
        self.acc = torch.zeros(4, dtype=torch.float64, device=self.device)

    def collect(self, env, reset=True):
        """Add (and by default clear) the statistics BppVecEnv accumulated inside its step kernels."""
        env.episode_stats(reset=reset, out=self.acc)      # the reduction kernel adds into self.acc itself
        return self

    def update(self, res):
        """Add the episodes that finished in step result `res` (StepTensors) with the stand-alone kernel
        (for callers that do not use the accumulator built into bpp_step)."""
        if self.device.type != "cuda":
            raise RuntimeError("EpisodeStats.update runs the HIP kernel; device tensors required")
        with torch.cuda.device(self.device):
            s = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.lib().bpp_episode_stats(res.done.data_ptr(), res.ep_ret.data_ptr(), res.ratio.data_ptr(),
                                                    res.ep_len.data_ptr(), res.done.numel(), self.acc.data_ptr(), s))

    def all_reduce(self, group=None):
        """Sum the record over all ranks (no-op without an initialised process group; with one the collective
        runs even for a single rank, so the RCCL path is the same code at every job size)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.acc, op=dist.ReduceOp.SUM, group=group)
        return self

    def summary(self):
        a = self.acc.cpu().tolist()
        n = max(a[3], 1.0)
        return {"episodes": int(a[3]), "mean_return": a[0] / n, "mean_ratio": a[1] / n, "mean_length": a[2] / n}

    def zero_(self):
        self.acc.zero_()
