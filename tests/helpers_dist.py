"""Test-only collectives for SEVERAL RANKS SHARING ONE GPU: RCCL needs a GPU per rank, gloo has no CUDA all-to-all, so the
exchange is staged through host memory.  The product (arithmetic-circuits_amd/parallel.py) only knows `Collectives` =
torch.distributed over RCCL; tests/dist_worker_gpu.py and bench.py's `--backend gloo` test mode pass this one instead."""
import importlib

import torch
import torch.distributed as dist

par = importlib.import_module("arithmetic-circuits_amd.parallel")


class HostStagedCollectives(par.Collectives):
    def all_reduce(self, t, op):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=self.group)
        t.copy_(c)

    def all_to_all(self, recv, send):
        torch.cuda.synchronize()
        sh = send.cpu()
        rh = torch.empty_like(sh)
        dist.all_to_all_single(rh, sh, group=self.group)
        recv.copy_(rh)
        return None                      # complete on return: nothing overlaps

    def overlaps(self, like):
        return False
