"""Clip prefetch for the CLI: the reference hides video decoding behind `DataLoader(..., num_workers=4)` (test/inference.py:341,348-351) -- four
worker processes read, decode, resize and pad the NEXT videos while the GPU runs the current one.

Here a clip's way to the GPU has two stages:

* HOST stage (`load_host`, thread pool): file read, container parse, JPEG decode of the frames the sampling schedule keeps, staging in PINNED host
  memory.  Pure host work in libraries that release the GIL (file IO, Pillow's libjpeg, numpy copies), so plain threads overlap it with the driver
  loop; `workers` of them, at most `depth` clips resident at once.
* DEVICE stage (the caller, main thread, stream-ordered): an asynchronous upload from the pinned buffer (+ the letterbox kernel for raw decoder output)
  -- issued for clip i + 1 on a copy stream right after clip i's frames were handed to the driver, so the PCIe transfer runs under clip i's LLM steps.

Results do not depend on any of this: the same functions produce the same frames, only earlier (tests/test_gpu_streams.py).
"""
import collections
from concurrent.futures import ThreadPoolExecutor
import torch


def pin(t):
    """Host tensor -> page-locked host tensor (a no-op without a GPU runtime: CPU-only test runs)."""
    if not torch.is_tensor(t) or t.device.type != 'cpu' or t.is_pinned() or not torch.cuda.is_available():
        return t
    try:
        return t.contiguous().pin_memory()
    except RuntimeError:                       # no usable GPU runtime behind torch.cuda (pinning is an optimisation, never a requirement)
        return t


class ClipPrefetcher:
    """`take()` returns `(index, load_host(index))` for `indices` IN ORDER (None at the end) while up to `depth` later clips are being loaded by
    `workers` threads; `next_ready()` says whether the next `take()` would return without waiting.  workers = 0 loads inline at `take()` (no prefetch)."""

    def __init__(self, load_host, indices, workers=4, depth=None, device=None):
        """device: the rank's GPU.  torch's current device is thread-local and starts at 0 in a new thread, so loader threads that pin memory would otherwise open a context
        (and allocate their page-locked buffers) on GPU 0 from every rank -- each loader thread selects the rank's device first, as torch's own DataLoader pin thread does."""
        self.load_host = load_host
        self.device = device
        self._todo = collections.deque(indices)
        self.workers = max(0, int(workers))
        self.depth = max(1, int(depth if depth is not None else max(1, self.workers)))
        self._pending = collections.deque()
        self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix='mmduet-clip', initializer=self._init_thread) if self.workers else None
        self._top_up()

    def _init_thread(self):
        dev = self.device
        if dev is not None and getattr(dev, 'type', None) == 'cuda' and torch.cuda.is_available():
            torch.cuda.set_device(dev)

    def _top_up(self):
        while self._pool is not None and self._todo and len(self._pending) < self.depth:
            i = self._todo.popleft()
            self._pending.append((i, self._pool.submit(self.load_host, i)))

    def next_ready(self):
        if self._pool is None:
            return False                       # inline loading always costs its time at take()
        return bool(self._pending) and self._pending[0][1].done()

    def exhausted(self):
        return not self._pending and not self._todo

    def take(self):
        if self._pool is None:
            if not self._todo:
                return None
            i = self._todo.popleft()
            return i, self.load_host(i)
        if not self._pending:
            return None
        i, fut = self._pending.popleft()
        self._top_up()                         # the slot this clip leaves is refilled before the consumer starts its (long) GPU work
        return i, fut.result()

    def __iter__(self):
        while True:
            r = self.take()
            if r is None:
                return
            yield r

    def close(self):
        if self._pool is not None:
            for _, f in self._pending:
                f.cancel()
            self._pool.shutdown(wait=True)
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
