"""HIP streams that really run concurrently.

Two HIP streams may be mapped to the same hardware queue, in which case their kernels serialise (measured: the fine-tune
step on two fresh streams ran 8 % SLOWER than on one, on three 20 % faster).  The mapping policy is the runtime's business,
so it is probed instead of assumed: a one-wave busy-wait kernel is launched on two candidate streams and the pair is
accepted when both finish in about the time of one."""
import ctypes as C
import time

import torch

_CACHE = {}


def _pair_is_concurrent(lib, a, b, cycles=3_000_000):
    """Device-side timing (events), and a busy-wait long enough (~1.5 ms) that the host's launch jitter cannot keep two
    concurrent-capable streams from overlapping."""
    def run(streams):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record(streams[0])
        ends = []
        for s in streams:
            s.wait_event(t0)
            lib.mi_debug_spin(cycles, C.c_void_p(s.cuda_stream))
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            ends.append(e)
        torch.cuda.synchronize()
        return max(t0.elapsed_time(e) for e in ends)
    run([a])
    one = min(run([a]) for _ in range(2))
    two = min(run([a, b]) for _ in range(2))
    return two < 1.5 * one


def concurrent_streams(n: int, device=None):
    """`n` streams of `device` that were measured to overlap pairwise (cached per device); falls back to whatever the pool
    offers when fewer independent hardware queues exist."""
    from . import _lib
    dev = torch.device(device if device is not None else "cuda", torch.cuda.current_device()) if not isinstance(device, torch.device) else device
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index
    chosen = _CACHE.setdefault(key, [])
    if len(chosen) >= n:
        return chosen[:n]
    lib = _lib.load()
    with torch.cuda.device(dev):
        pool = [torch.cuda.Stream(device=dev) for _ in range(12)]
        for cand in pool:
            if len(chosen) >= n:
                break
            if all(_pair_is_concurrent(lib, c, cand) for c in chosen):
                chosen.append(cand)
        k = 0
        while len(chosen) < n:  # not enough independent queues: reuse
            chosen.append(pool[k % len(pool)])
            k += 1
    return chosen[:n]


class ChainWorkers:
    """Long-lived host threads, one per concurrent stream, that enqueue the chains of `sample()` calls.

    A sampler call used to start (and join) one Python thread per crystal group: 1.3-1.7 ms of thread start and state set-up per call
    and a ragged start of the chains -- 3-6 % of a 20-step window.  The workers are created once per device and process,
    each bound to ITS stream of `concurrent_streams`; a call hands every worker a closure and waits for all of them.  The work itself is
    unchanged (nothing moves out of a timed region: the start-up cost is removed, not hidden).  Exceptions are re-raised on the caller's
    thread; daemon threads, so an interpreter exit never waits for them."""

    _pools = {}
    _lock = None

    def __init__(self, n, device):
        import queue
        import threading
        self.streams = concurrent_streams(n, device)
        self.device = self.streams[0].device
        self._q = [queue.SimpleQueue() for _ in range(n)]
        self._busy = threading.Lock()   # one sample() call at a time per pool (the chains of two calls would share streams)
        self._threads = [threading.Thread(target=self._loop, args=(k,), name=f"mi-chain-{k}", daemon=True) for k in range(n)]
        for t in self._threads:
            t.start()

    def _loop(self, k):
        torch.cuda.set_device(self.device)
        while True:
            item = self._q[k].get()
            if item is None:
                return
            fn, done = item   # (every call brings its OWN result queue: what an interrupted call left behind can never reach a later one)
            try:
                with torch.cuda.stream(self.streams[k]):
                    done.put((k, fn(k, self.streams[k]), None))
            except BaseException as e:  # noqa: BLE001 -- handed to the caller
                done.put((k, None, e))

    def run(self, fn, n=None):
        """fn(k, stream) on worker k = 0 .. n-1 (each under its stream); returns the list of results in worker order."""
        import queue
        n = len(self._q) if n is None else n
        done = queue.SimpleQueue()
        with self._busy:
            for k in range(n):
                self._q[k].put((fn, done))
            out, err, got = [None] * n, None, 0
            try:
                while got < n:
                    k, r, e = done.get()
                    got += 1
                    out[k] = r
                    err = err or e
            finally:
                # interrupted while waiting (KeyboardInterrupt in a notebook): the workers are still enqueueing on their streams.  Wait for
                # them before the pool is released, so the next call neither shares a stream with a half-enqueued chain nor sees its results.
                while got < n:
                    done.get()
                    got += 1
        if err is not None:
            raise err
        return out

    @classmethod
    def get(cls, n, device=None):
        import threading
        if cls._lock is None:
            cls._lock = threading.Lock()
        dev = torch.device(device if device is not None else "cuda") if not isinstance(device, torch.device) else device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with cls._lock:
            pool = cls._pools.get(idx)
            if pool is None or len(pool._q) < n:
                if pool is not None:
                    for q in pool._q:
                        q.put(None)
                pool = cls(max(n, 4), torch.device("cuda", idx))
                cls._pools[idx] = pool
        return pool
