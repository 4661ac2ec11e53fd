"""Data-parallel streaming over the GPUs of one node: independent videos shard across ranks, scores are gathered.

The reference has no inference-time parallelism; its seam is `--start_idx/--end_idx` (models/arguments_live.py:50-51,
test/inference.py:337) for N manually launched processes.  Here one process per GPU (torch.distributed, backend
'nccl' = RCCL over xGMI) takes streams `i % world == rank`; the only collective on the path is ONE all-gather of the
per-frame head scores ([T,2] fp32 per stream, KB-scale, latency-bound -- no ring tuning applies).  Each video stream is
an independent recurrence over its own KV arena, so there is no other exchange step.

Two equivalent transports:
  * `gather_scores`       torch.distributed (`all_gather_into_tensor`), any backend (gloo on CPU for the tests);
  * `NativeScoreGather`   the C-ABI entry `mmd_gather_scores` (include/mmduet.h): one ncclAllGather issued by libmmduet_hip
                          itself on the model's stream, the communicator's id handed around through the launcher's store.
"""
import ctypes as C
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


def shard_shape(n_items, world, lengths=None):
    """(n_max, t_max) of the padded score block, computed WITHOUT communication: the assignment is deterministic, so every rank
    knows how many streams the busiest rank holds and (with `lengths`) the longest stream."""
    n_max = max((len(shard_indices(n_items, r, world, lengths)) for r in range(world)), default=0)
    t_max = max(lengths, default=0) if lengths is not None else None
    return n_max, t_max


def gather_scores(local_scores, t_max=None, n_max=None):
    """All-gather per-stream score arrays in ONE collective.

    local_scores: list of float tensors [T_i, 2] (informative, relevance) for the streams of this rank (any device).
    Returns (scores [world, n_max, t_max, 2] fp32 padded with NaN, lengths [world, n_max] int32) on every rank; with
    world == 1 no collective is issued.

    The block every rank contributes is [n_max, t_max + 1, 2]: row 0 of a stream carries (T_i, 0) -- the length travels inside
    the score block, so lengths need no collective of their own.  `n_max` / `t_max` are known to every rank without
    communication when the dataset is (`shard_shape`); only if they are not given does a second, 16-byte all-gather of the
    local shape precede the score gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    dev = torch.device('cuda', torch.cuda.current_device()) if (dist.is_initialized() and dist.get_backend() == 'nccl') else torch.device('cpu')
    n_local = len(local_scores)
    t_local = max([int(s.shape[0]) for s in local_scores], default=0)
    if world > 1 and (t_max is None or n_max is None):
        meta = torch.tensor([n_local, t_local], dtype=torch.int64, device=dev)
        metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta)
        metas = metas.view(world, 2).cpu()
        n_max = int(metas[:, 0].max()) if n_max is None else n_max
        t_max = int(metas[:, 1].max()) if t_max is None else t_max
    n_max = n_local if n_max is None else int(n_max)
    t_max = t_local if t_max is None else int(t_max)
    if n_local > n_max or t_local > t_max:
        raise ValueError(f'local block [{n_local},{t_local}] exceeds the agreed [{n_max},{t_max}]')
    buf = torch.full((n_max, t_max + 1, 2), float('nan'), dtype=torch.float32)
    buf[:, 0] = 0.0
    for i, s in enumerate(local_scores):
        buf[i, 0, 0] = float(s.shape[0])
        buf[i, 1:1 + s.shape[0]] = s.detach().to(device='cpu', dtype=torch.float32)
    buf = buf.to(dev)
    if world == 1:
        out = buf[None]
    else:
        out = torch.empty((world, n_max, t_max + 1, 2), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out.view(world * n_max, t_max + 1, 2), buf)          # the ONE collective
    return out[:, :, 1:], out[:, :, 0, 0].to(torch.int32)


def gather_responses(local_ids, n_max):
    """Response mode: the generated token ids of every stream travel with the scores ("for response mode also [n_resp, 1 + L] int64 token ids").
    local_ids: per stream of this rank a list of responses, each a list of token ids.  Returns a list [world][n_max] of lists of id lists (empty beyond a rank's streams).
    One 16-byte shape exchange (responses per stream and their length are not known in advance), then ONE all-gather of the padded block [n_max, r_max, 1 + l_max] int64
    (column 0 = the response's length, -1 = no response) -- skipped on every rank alike when no rank generated anything."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    dev = torch.device('cuda', torch.cuda.current_device()) if (dist.is_initialized() and dist.get_backend() == 'nccl') else torch.device('cpu')
    r_loc = max([len(r) for r in local_ids], default=0)
    l_loc = max([len(t) for r in local_ids for t in r], default=0)
    if len(local_ids) > n_max:
        raise ValueError('more local streams than the agreed n_max')
    r_max, l_max = r_loc, l_loc
    if world > 1:
        meta = torch.tensor([r_loc, l_loc], dtype=torch.int64, device=dev)
        metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta)
        metas = metas.view(world, 2).cpu()
        r_max, l_max = int(metas[:, 0].max()), int(metas[:, 1].max())
    if r_max == 0:
        return [[[] for _ in range(n_max)] for _ in range(world)]
    buf = torch.full((n_max, r_max, 1 + l_max), -1, dtype=torch.int64)
    for i, resp in enumerate(local_ids):
        for j, t in enumerate(resp):
            buf[i, j, 0] = len(t)
            buf[i, j, 1:1 + len(t)] = torch.as_tensor(t, dtype=torch.int64)
    buf = buf.to(dev)
    if world == 1:
        out = buf[None]
    else:
        out = torch.empty((world, n_max, r_max, 1 + l_max), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out.view(world * n_max, r_max, 1 + l_max), buf)
    out = out.cpu()
    return [[[out[w, i, j, 1:1 + int(out[w, i, j, 0])].tolist() for j in range(r_max) if int(out[w, i, j, 0]) >= 0] for i in range(n_max)] for w in range(world)]


class NativeScoreGather:
    """`mmd_gather_scores` (include/mmduet.h): the all-gather issued by libmmduet_hip itself (RCCL bound at run time) on the
    model's stream.  The 128-byte communicator id is drawn on rank 0 and handed to the other ranks through a torch.distributed
    broadcast (construction only); the data path never touches torch.distributed."""

    def __init__(self, device, rank=None, world=None, stream=None):
        from ._lib import lib, MmduetError
        self._lib, self._err = lib(), MmduetError
        self.rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        self.world = dist.get_world_size() if world is None and dist.is_initialized() else (world or 1)
        self.device = torch.device(device)
        self._h = None
        # Construction is collective (broadcast + ncclCommInitRank): first agree that EVERY rank can bind librccl and that rank 0 drew an id -- a rank that
        # cannot must make all of them raise here, not leave the others waiting inside ncclCommInitRank.
        ok = 1 if self._lib.mmd_comm_probe() == 0 else 0
        ident = torch.zeros(128, dtype=torch.uint8)
        if ok and self.rank == 0 and self._lib.mmd_comm_unique_id(C.c_void_p(ident.data_ptr())) != 0:
            ok = 0
        why = '' if ok else self._lib.mmd_comm_last_error(None).decode()
        if self.world > 1:
            nccl = dist.get_backend() == 'nccl'
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device if nccl else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                raise MmduetError(f'NativeScoreGather: RCCL is not usable on every rank (rank {self.rank}: {why or "ok here"})')
            on = ident.to(self.device) if nccl else ident
            dist.broadcast(on, src=0)
            ident = on.cpu()
        elif not ok:
            raise MmduetError(f'NativeScoreGather: {why}')
        self._id = ident
        h = C.c_void_p()
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.mmd_comm_create(C.c_void_p(ident.data_ptr()), self.rank, self.world, self.device.index or 0, C.c_void_p(st), C.byref(h))
        why = '' if rc == 0 else f'mmd_comm_create failed ({rc}): {self._lib.mmd_comm_last_error(None).decode()}'
        if self.world > 1:
            # ncclCommInitRank can fail on SOME ranks only: agree once more, so that either every rank holds a communicator or every rank raises (a caller that
            # falls back to torch.distributed then does so on all ranks together -- mixed transports would hang on mismatched collectives)
            flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=self.device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if rc == 0:
                    self._lib.mmd_comm_destroy(h)
                raise MmduetError(f'NativeScoreGather: communicator construction failed on at least one rank (rank {self.rank}: {why or "ok here"})')
        elif rc:
            raise MmduetError(why)
        self._h = h

    def _on_current_stream(self):
        """Every gather is issued on torch's CURRENT stream: its input was produced there and its output is consumed there, so stream order is the only
        ordering needed (a communicator bound to the stream of its construction would race with work on any other stream)."""
        rc = self._lib.mmd_comm_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc:
            raise self._err(f'mmd_comm_set_stream failed ({rc}): {self._lib.mmd_comm_last_error(self._h).decode()}')

    def gather(self, scores, t_max):
        """scores [T,2] fp32 on the device -> (all [world, t_max, 2] fp32 NaN-padded, lengths [world] int32), device tensors."""
        s = scores.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty(self.world, t_max + 1, 2, dtype=torch.float32, device=self.device)
        self._on_current_stream()
        rc = self._lib.mmd_gather_scores(self._h, C.c_void_p(s.data_ptr()), int(s.shape[0]), int(t_max), C.c_void_p(out.data_ptr()))
        if rc:
            raise self._err(f'mmd_gather_scores failed ({rc}): {self._lib.mmd_comm_last_error(self._h).decode()}')
        return out[:, 1:], out[:, 0, 0].to(torch.int32)

    def gather_streams(self, local_scores, t_max, n_max):
        """Several streams per rank, same result layout as `gather_scores`: list of [T_i,2] -> (scores [world, n_max, t_max, 2] NaN-padded, lengths [world, n_max] int32).
        The padded block [n_max, t_max + 1, 2] (row 0 of a stream = its length) is assembled once on the host -- the scores are host values, the driver decided on
        them frame by frame -- and crosses to the device in ONE copy; ONE ncclAllGather issued by the library."""
        if len(local_scores) > n_max or any(int(s.shape[0]) > t_max for s in local_scores):
            raise ValueError('local block exceeds the agreed [n_max, t_max]')
        buf = torch.full((n_max, t_max + 1, 2), float('nan'), dtype=torch.float32)
        buf[:, 0] = 0.0
        for i, s in enumerate(local_scores):
            buf[i, 0, 0] = float(s.shape[0])
            buf[i, 1:1 + s.shape[0]] = s.detach().to(device='cpu', dtype=torch.float32)
        blk = buf.to(self.device, non_blocking=True)
        out = torch.empty(self.world, n_max, t_max + 1, 2, dtype=torch.float32, device=self.device)
        self._on_current_stream()
        rc = self._lib.mmd_gather_block(self._h, C.c_void_p(blk.data_ptr()), blk.numel(), C.c_void_p(out.data_ptr()))
        if rc:
            raise self._err(f'mmd_gather_block failed ({rc}): {self._lib.mmd_comm_last_error(self._h).decode()}')
        return out[:, :, 1:], out[:, :, 0, 0].to(torch.int32)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.mmd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
