"""Ensemble sharding over the GPUs of one node: one process per GPU, a contiguous block of
members per rank, no data-path collective.  The only exchange is the ensemble-mean of
(time-averaged) spectra at diagnostic time, the reference's ``ds[spec].mean('run')``
(pyqg_generative/tools/comparison_tools.py:167-168,371; tools/simulate.py:284-290),
done as ONE all-reduce of per-rank partial sums over RCCL ("nccl" backend on ROCm) or gloo.
"""
import os
import torch


def rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def shard_members(total_members, rank, world):
    """Contiguous, balanced blocks: -> (first_member, n_local).  sum(n_local) == total."""
    if not (0 <= rank < world) or total_members < 0:
        raise ValueError('bad rank/world/total')
    base, rem = divmod(total_members, world)
    n_local = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n_local


def init_process_group(backend=None, single_rank=False):
    """torch.distributed bootstrap from the torchrun environment (127.0.0.1 rendezvous).  A one-rank job needs no
    process group (-> None) unless ``single_rank``: then the group is created all the same, so that the collective
    path — RCCL on a one-GPU box — can be exercised (tests/test_gpu_facade.py)."""
    import torch.distributed as dist
    rank, world = rank_world()
    if world == 1 and not single_rank:
        return None
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {}
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', rank))
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device('cuda', local)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def ensemble_mean(local_sum, local_count, group=None):
    """Mean over ALL members of the job of a quantity whose per-rank SUM over local members is
    ``local_sum`` (tensor, e.g. (2,N,NK) float64 spectra).  One all-reduce of numel+1 doubles."""
    import torch.distributed as dist
    buf = torch.cat([local_sum.reshape(-1).to(torch.float64),
                     torch.tensor([float(local_count)], dtype=torch.float64, device=local_sum.device)])
    if dist.is_available() and dist.is_initialized():          # (a one-rank group too: the same code path at every job size)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (buf[:-1] / buf[-1]).reshape(local_sum.shape)
