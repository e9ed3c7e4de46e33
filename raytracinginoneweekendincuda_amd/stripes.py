"""The one exchange step of a multi-GPU frame: every rank renders its 8-row stripes into a compact buffer, one gather
brings the buffers to rank 0, rank 0 de-interleaves them into the full frame (the reference has no
multi-GPU path -- its frame is one `Vector3 frameBuffer[W * H]`, R/kernel.cu:606-609 -- pixels are independent because the
RNG sequence of a pixel is its global index, R/kernel.cu:117-118).

Backend-agnostic on purpose: `dist` is torch.distributed with whatever backend the caller initialised ("nccl" = RCCL over
xGMI on the GPUs, "gloo" on CPU in the tests), `device` is where the rank's buffer lives.  bench.py and
tests/test_dist_gloo.py both go through this class, so the CPU test exercises the code the GPU run uses.
"""
import numpy as np

from . import api


class StripeExchange:
    def __init__(self, dist, width, height, stripe_rows, rank, world_size, device):
        import torch
        self.dist, self.rank, self.world = dist, rank, world_size
        self.width, self.height, self.stripe = width, height, stripe_rows
        self.rows = api.stripe_rows(height, stripe_rows, rank, world_size)            # rows of the full frame this rank owns
        self.rows_max = max(len(api.stripe_rows(height, stripe_rows, r, world_size)) for r in range(world_size))
        # every rank's buffer has the size of the largest share: gather wants equal tensors
        self.mine = torch.zeros(self.rows_max * width * 3, dtype=torch.float64, device=device)
        self.gathered = ([torch.empty_like(self.mine) for _ in range(world_size)]
                         if (world_size > 1 and rank == 0) else None)

    def gather(self):
        """The frame's single collective (nothing to do on one rank)."""
        if self.world > 1:
            self.dist.gather(self.mine, self.gathered, dst=0)

    def frame(self):
        """Rank 0: the full H x W x 3 frame from what the last gather brought (rank-major compact buffers)."""
        import torch
        if self.rank != 0:
            return None
        parts = self.gathered if self.world > 1 else [self.mine]
        stacked = torch.stack([p.detach().cpu() for p in parts]).numpy()
        return api.deinterleave(np.ascontiguousarray(stacked), self.width, self.height, self.stripe, self.world)
