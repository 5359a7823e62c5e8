"""The random side of the SO(N) generator (optex.py:142-149): the numpy-legacy gaussian stream that scipy's
special_ortho_group.rvs consumes.  Drawing from numpy's GLOBAL RandomState (the default) reproduces the reference's
matrices after np.random.seed(s); the O(N^3) Householder accumulation runs on the GPU (csrc/rotation.hip).  The stream
itself can run on the GPU too (DeviceNormals): same MT19937 words, same polar method and cache, advanced by
optex_legacy_normals from numpy's own state tuple."""
import os
import warnings
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib, ops

_pool = None


def draw_normals(N: int, count: int = 1, rng=None) -> np.ndarray:
    """[count, N(N+1)/2 - 1] float64: per rotation, the concatenated normal(size=N-n) draws for n = 0..N-2.
    RandomState.normal is a sequential stream (the one-value gaussian cache carries over), so one big draw equals
    scipy's N-1 small ones."""
    if N is None or not np.isscalar(N) or N <= 1 or N != int(N):
        raise ValueError("Dimension of rotation must be specified,\n and must be a scalar greater than 1.")
    per = ops.rotation_normals(int(N))
    src = np.random if rng is None else rng
    return src.normal(size=(count, per))


def rotations(N: int, count: int, device, rng=None, want64=False):
    """count Haar-random rotations as device tensors: (R32 [count,N,N], Rt32 transposes[, R64])"""
    return ops.rotations_from_normals(draw_normals(N, count, rng), int(N), count, device, want64=want64)


def rotations_per_segment(N: int, count: int, device, rngs):
    """One rotation sequence PER SEGMENT (the reference run once per image draws its own rotations, optex.py:149,168):
    rngs = one numpy RandomState per segment, segment s gets the `count` rotations its stream yields — exactly what a
    B = 1 run seeded like rngs[s] would draw.  Returns (R32 [S, count, N, N], Rt32).  The streams are sequential by
    construction (MT19937 + polar method with a one-value cache), so the host draw is the cost of this mode: S * count *
    (N (N + 1) / 2 - 1) normals, drawn on a thread pool (numpy's legacy generator releases the GIL)."""
    global _pool
    S = len(rngs)
    per = ops.rotation_normals(int(N))
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=max(1, min(64, os.cpu_count() or 1)))
    normals = np.empty((S, count, per), dtype=np.float64)

    def draw(i):
        normals[i] = rngs[i].normal(size=(count, per))

    list(_pool.map(draw, range(S)))
    R32, Rt32 = ops.rotations_from_normals(normals.reshape(S * count, per), int(N), S * count, device)
    return R32.view(S, count, N, N), Rt32.view(S, count, N, N)


_side_streams = {}


def _generator_stream(device):
    """ONE high-priority side stream per device for all generators (bench.py makes a DeviceNormals per step: a stream each would
    be a hipStreamCreate / Destroy per step); the draws of different generators simply queue behind each other on it"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(torch.device("cuda", key), priority=-1)
    return _side_streams[key]


class DeviceNormals:
    """numpy RandomState gaussian streams that live on the GPU (csrc/rotation.hip mt_accept_kernel + normals_emit_kernel,
    optex_legacy_normals, ABI 7).

    Built from numpy RandomState objects (their get_state() tuples are copied to the device once) or from integer seeds
    (seeded on the device); from then on every draw is a pair of kernels — one workgroup per stream walks the MT19937 words
    and the accept / reject decisions, all CUs do the arithmetic — and the host never touches a normal again: no 17 ms of
    RandomState.normal and no 24 ms of pageable host-to-device copies per 52-iteration step (scripts/host_profile.py), and
    64 per-texture streams advance side by side instead of on a host thread pool.  Words, decisions and state are numpy's
    exactly; >= 99.8 % of the values equal RandomState.normal's bit for bit, the rest differ by <= 4 ulp (the device's log is
    correctly rounded, glibc's is not quite).

    One stream: rotations(N, count) -> (R32 [count, N, N], Rt32), the whole batch shares the sequence like the reference's
    --batch (optex.py:168-170).  S streams: -> ([S, count, N, N], ...), one sequence per texture (the reference run once
    per image).  The kernels run on a side stream; prefetch(schedule) enqueues the draws of a whole forward call up front
    (they are sequentially dependent through the state, but independent of the features), so they overlap the
    convolutions and the OT loops find their rotations ready."""

    def __init__(self, rngs, device, side_stream: bool = True):
        """rngs: one or a list of numpy RandomState objects (their current state is taken over) or 32-bit integer seeds
        (RandomState(seed), seeded on the device: nothing crosses the bus)"""
        rngs = list(rngs) if isinstance(rngs, (list, tuple)) else [rngs]
        lib = _lib.lib()
        words = lib.optex_mt19937_state_bytes() // 4
        self.device = torch.device(device)
        self.n = len(rngs)
        # (a high-priority stream: the generator's one workgroup per stream should not queue behind the main stream's launches)
        self.stream = _generator_stream(self.device) if side_stream else None
        self._queue = deque()   # prefetched rotations in stream order: (N, count, (R32, Rt32), event)
        run = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        ints = all(isinstance(r, (int, np.integer)) and 0 <= int(r) < 2 ** 32 for r in rngs)
        stride = (int(rngs[1]) - int(rngs[0])) % 2 ** 32 if ints and self.n > 1 else 0
        with torch.cuda.stream(run):
            self.states = torch.empty((self.n, words), dtype=torch.int32, device=self.device)
            if ints and all((int(rngs[0]) + i * stride) % 2 ** 32 == int(r) for i, r in enumerate(rngs)):
                _lib.check(lib.optex_mt19937_seed(_lib.ptr(self.states), self.n, int(rngs[0]), stride, ctypes_stream(run)))
                return
            host = torch.empty((self.n, words), dtype=torch.int32).pin_memory()   # (torch caches pinned blocks)
            hv = host.numpy().view(np.uint32)
            for i, r in enumerate(rngs):
                if not isinstance(r, np.random.RandomState):
                    r = np.random.RandomState(int(r))
                name, key, pos, has_gauss, cached = r.get_state()
                assert name == "MT19937" and key.shape == (624,)
                hv[i, :624] = key
                hv[i, 624] = pos
                hv[i, 625] = has_gauss
                hv[i, 626:628] = np.frombuffer(np.float64(cached).tobytes(), dtype=np.uint32)
            self.states.copy_(host, non_blocking=True)   # asynchronous, on the generator's stream: the host does not wait

    def state(self, i: int = 0):
        """the stream's state as numpy's get_state() tuple (a host synchronisation: tests, hand-over back to the host)"""
        if self.stream is not None:
            self.stream.synchronize()
        w = self.states[i].cpu().numpy().view(np.uint32)
        return ("MT19937", w[:624].copy(), int(w[624]), int(w[625]), float(np.frombuffer(w[626:628].tobytes(), dtype=np.float64)[0]))

    def draw(self, total: int):
        """the next `total` values of every stream: ([n, total] float64 on the device, event recorded behind the draw or None)"""
        lib = _lib.lib()
        cur = torch.cuda.current_stream(self.device)
        run = self.stream if self.stream is not None else cur
        with torch.cuda.stream(run):
            out = torch.empty((self.n, int(total)), dtype=torch.float64, device=self.device)
            ws = _lib.workspace(lib.optex_legacy_normals_ws_bytes(self.n, int(total)), self.device)
            _lib.check(lib.optex_legacy_normals(_lib.ptr(self.states), self.n, int(total), _lib.ptr(out), int(total),
                                                _lib.ptr(ws), ws.numel(), ctypes_stream(run)))
            ev = None
            if self.stream is not None:
                ev = torch.cuda.Event()
                ev.record(run)
        return out, ev

    def _rotations_from(self, normals, N, count):
        R32, Rt32 = ops.rotations_from_normals(normals.view(self.n * count, -1), N, self.n * count, self.device)
        if self.n == 1:
            return R32, Rt32
        return R32.view(self.n, count, N, N), Rt32.view(self.n, count, N, N)

    # prefetched rotations kept alive at once: beyond this the remaining (pass, layer) entries of a schedule are
    # drawn when they are asked for (still on the generator's stream, still in stream order)
    PREFETCH_BYTES = 2 << 30

    def pending(self):
        """[(N, count), ...] of the prefetched rotations nobody has taken yet, in stream order"""
        return [(n, c) for n, c, _, _ in self._queue]

    def covers(self, schedule) -> bool:
        """the pending prefetch is exactly the head of `schedule` (non-empty entries): a forward() call that finds its
        schedule covered does not draw again — the draws were enqueued ahead of time, e.g. during the previous step"""
        want = [(int(n), int(c)) for n, c in schedule if c > 0]
        have = self.pending() + list(getattr(self, "_feed", []) or [])
        # ... and it was enqueued FOR this schedule (prefetch / begin_feed tag their queue): the leftover tail of another schedule
        # that happens to equal this one's head is not a cover — forward() then drops it (counted, warned) and draws
        return len(have) > 0 and have == want[:len(have)] and getattr(self, "_tag", None) == tuple(want)

    def drop_pending(self):
        """forget prefetched rotations (a forward() that raised midway, a schedule that changed): the stream stays where the
        draws left it — the dropped values are consumed, exactly as if someone had asked for them and thrown them away"""
        n_drawn = sum(int(c) for _, c, _, _ in self._queue)
        if n_drawn:
            # the numpy-compatible stream has moved past these draws: every later rotation differs from what the reference would
            # draw for the same seed.  Counted, and said once per generator.
            self.dropped_rotations = getattr(self, "dropped_rotations", 0) + n_drawn
            if not getattr(self, "_warned_drop", False):
                self._warned_drop = True
                warnings.warn(f"DeviceNormals: {n_drawn} rotation(s) drawn ahead were discarded because the request sequence left the "
                              "prefetched schedule; the gaussian stream has consumed them, so later rotations differ from the "
                              "reference's sequence for this seed (further drops are counted in .dropped_rotations)",
                              RuntimeWarning, stacklevel=3)
        self._queue.clear()
        self._feed = []
        self._tag = None

    # ---- a schedule fed piece by piece: the draws of the NEXT call released one (pass, layer) at a time by the CURRENT call, each
    # at the start of one of its VGG codec phases (driver.OptimalTexture.forward, `rng_next`).  All at once (prefetch) the
    # generator's ~12 ms of one compute unit run beside whatever the main stream does then — at 8 textures per step that is the
    # first OT loops, whose persistent rotation GEMM wants every CU whole (include/optex.h, optex_gemm_spare_cus).  The
    # convolutions do not mind sharing a CU with the generator's one workgroup, the GEMM does.
    def begin_feed(self, schedule):
        """start a fed schedule: nothing is enqueued yet; feed_one() releases the entries in order, finish_feed() the rest"""
        self.drop_pending()
        self._feed = [(int(n), int(c)) for n, c in schedule if c > 0]
        self._tag = tuple(self._feed)
        self._fed_bytes = 0

    def feeding(self) -> bool:
        return bool(getattr(self, "_feed", None))

    def feed_one(self, after=None):
        """enqueue the next entry of the fed schedule on the generator's stream, behind `after` (an event of the caller's
        stream: the point in ITS timeline from which the draw may run).  Returns False when nothing is left to feed."""
        if not self.feeding():
            return False
        N, count = self._feed[0]
        self._fed_bytes += 8 * self.n * count * N * N
        if self._fed_bytes > self.PREFETCH_BYTES:   # (the rest is drawn when it is asked for, like prefetch() does)
            self._feed = []
            return False
        self._feed.pop(0)
        if self.stream is not None and after is not None:
            self.stream.wait_event(after)
        self._enqueue(N, count)
        return True

    def finish_feed(self):
        """enqueue whatever of the fed schedule has not been released yet (a call with fewer codec phases than the next one
        has entries, a caller that stops feeding)"""
        while self.feed_one():
            pass

    def _enqueue(self, N, count):
        normals, ev = self.draw(count * ops.rotation_normals(N))
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                R = self._rotations_from(normals, N, count)
                ev = torch.cuda.Event()
                ev.record(self.stream)
        else:
            R = self._rotations_from(normals, N, count)
        self._queue.append((N, count, R, ev))

    def prefetch(self, schedule):
        """schedule: [(N, count), ...] in the order the rotations will be asked for.  Draws AND Householder accumulations go
        out on the generator's stream now (with one sequence per texture the accumulations of a bench step are 3 328
        rotations of 256^2, 27 ms that would otherwise sit between the convolutions and every OT loop).  Leftovers of an
        earlier schedule are dropped first; at most PREFETCH_BYTES of rotations are kept ahead."""
        self.drop_pending()
        self._tag = tuple((int(n), int(c)) for n, c in schedule if c > 0)
        held = 0
        for N, count in schedule:
            if count > 0:
                N, count = int(N), int(count)
                held += 8 * self.n * count * N * N
                if held > self.PREFETCH_BYTES:
                    break
                self._enqueue(N, count)

    def rotations(self, N: int, count: int):
        if N is None or not np.isscalar(N) or N <= 1 or N != int(N):
            raise ValueError("Dimension of rotation must be specified,\n and must be a scalar greater than 1.")
        N, count = int(N), int(count)
        cur = torch.cuda.current_stream(self.device)
        if not self._queue and self.feeding():
            self.feed_one()   # asked for before its release point came: draw it now (same stream position, same values)
        if self._queue:
            qn, qc, R, ev = self._queue[0]
            if (qn, qc) == (N, count):
                self._queue.popleft()
                if ev is not None:
                    cur.wait_event(ev)
                    for t in R:
                        t.record_stream(cur)
                return R
            # the caller left the prefetched schedule (another k after a re-fit, a call that raised midway): what was
            # drawn ahead is dropped — consumed, like values asked for and thrown away — and this request is drawn now
            self.drop_pending()
        normals, ev = self.draw(count * ops.rotation_normals(N))
        if ev is not None:
            cur.wait_event(ev)
            normals.record_stream(cur)
        return self._rotations_from(normals, N, count)


def ctypes_stream(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)
