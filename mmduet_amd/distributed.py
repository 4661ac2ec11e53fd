"""Data-parallel streaming over the GPUs of one node: independent videos shard across ranks, scores are gathered.

The reference has no inference-time parallelism; its seam is `--start_idx/--end_idx` (models/arguments_live.py:50-51,
test/inference.py:337) for N manually launched processes.  Here one process per GPU (torch.distributed, backend
'nccl' = RCCL over xGMI) takes streams `i % world == rank`; the only collective on the path is ONE all-gather of the
per-frame head scores ([T,2] fp32 per stream, KB-scale, latency-bound -- no ring tuning applies).  Each video stream is
an independent recurrence over its own KV arena, so there is no other exchange step.
"""
import os
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items, rank, world, lengths=None):
    """Indices of the streams this rank processes.  Round-robin by index; with `lengths`, longest-first greedy
    balancing (deterministic on every rank)."""
    if lengths is None:
        return [i for i in range(n_items) if i % world == rank]
    order = sorted(range(n_items), key=lambda i: (-lengths[i], i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        load[r] += lengths[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_scores(local_scores, t_max=None):
    """All-gather per-stream score arrays.

    local_scores: list of float tensors [T_i, 2] (informative, relevance) for the streams of this rank (any device).
    Returns (scores [world, n_max, t_max, 2] fp32 padded with NaN, lengths [world, n_max] int32) on every rank; with
    world == 1 no collective is issued."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    dev = torch.device('cuda', torch.cuda.current_device()) if (dist.is_initialized() and dist.get_backend() == 'nccl') else torch.device('cpu')
    n_local = len(local_scores)
    t_local = max([int(s.shape[0]) for s in local_scores], default=0)
    meta = torch.tensor([n_local, t_local if t_max is None else t_max], dtype=torch.int64, device=dev)
    if world > 1:
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta)
        n_max = max(int(m[0]) for m in metas)
        t_max = max(int(m[1]) for m in metas)
    else:
        n_max, t_max = n_local, int(meta[1])
    buf = torch.full((n_max, t_max, 2), float('nan'), dtype=torch.float32, device=dev)
    lens = torch.zeros(n_max, dtype=torch.int32, device=dev)
    for i, s in enumerate(local_scores):
        buf[i, :s.shape[0]] = s.to(device=dev, dtype=torch.float32)
        lens[i] = s.shape[0]
    if world == 1:
        return buf[None], lens[None]
    out = torch.empty((world * n_max, t_max, 2), dtype=torch.float32, device=dev)
    out_l = torch.empty((world * n_max,), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, buf)          # one RCCL all-gather of the padded score block (concatenated along dim 0)
    dist.all_gather_into_tensor(out_l, lens)
    return out.view(world, n_max, t_max, 2), out_l.view(world, n_max)
