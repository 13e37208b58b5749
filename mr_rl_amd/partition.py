"""CuPartition: the GPU's compute units split between the collection launches and the learner, so that the two halves of the
DDPG loop (RL/MR_ddpg.py:270-311: act / step / store, then the update) overlap on ONE device.

Why a partition.  A fused-rollout launch group of 262 144 envs is exactly one resident round of workgroups on 256 units.  Work
enqueued meanwhile on an ordinary stream -- the learner's one-workgroup update burst (152 KB of LDS: a whole unit), the replay
push, the parameter upload the next-but-one episode waits for -- finds a unit only at a launch boundary, where it races the next
launch group for it, and then holds up the collection blocks it displaced for its own duration.  Measured without a partition:
the loop runs at 406 us per episode against 272 us of collection kernel with NO updates at all, and every update adds its full
49 us (profiles/r04/NOTES.md).

The partition: hipExtStreamCreateWithCUMask (through libmrsim: mrsim_stream_create_cu_mask).  Mask bit i is a unit of XCC i % 8
(tools/cumask_probe.hip); workgroups are dealt round robin over the XCCs whatever the mask says, so both sides keep units in
every XCC: the learner gets `per_xcc` units of each XCC (8 units for per_xcc = 1: 3 % of the device), the collection streams all
the others.  The collection launches no longer fit one round; with the env set cut into 8 or 16 sub-shards on as many streams the
blocks that do not fit run beside the next launch group of the other sub-shards (no env waits for an env of another sub-shard).
"""
import ctypes as C

from . import _lib


def partition_masks(compute_units, xccs, per_xcc):
    """(learner mask, collection mask) as lists of 32-bit words: the learner gets bits 0 .. per_xcc * xccs - 1 (mask bit i is a unit
    of XCC i % xccs, so that is per_xcc units of every XCC), the collection every other unit.  Both sides keep a unit in every XCC."""
    compute_units, xccs, per_xcc = int(compute_units), int(xccs), int(per_xcc)
    if per_xcc < 1 or per_xcc * xccs * 2 > compute_units:
        raise ValueError("per_xcc must leave the collection at least half of the device")
    n_learner = per_xcc * xccs
    words = (compute_units + 31) // 32
    learner, collect = [0] * words, [0] * words
    for b in range(compute_units):
        (learner if b < n_learner else collect)[b // 32] |= 1 << (b % 32)
    return learner, collect


class CuPartition:
    def __init__(self, device="cuda", per_xcc=1, collection_streams=8):
        import torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CuPartition needs a HIP device (libmrsim has no CPU fallback)")
        self._L = _lib.lib()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.index = int(idx)
        cus, xccs = C.c_int32(0), C.c_int32(0)
        _lib.check(self._L.mrsim_device_cu_layout(self.index, C.byref(cus), C.byref(xccs)), "mrsim_device_cu_layout")
        self.compute_units, self.xccs = int(cus.value), int(xccs.value)
        learner, collect = partition_masks(self.compute_units, self.xccs, per_xcc)
        self.learner_units = int(per_xcc) * self.xccs
        self.learner_mask, self.collection_mask = learner, collect
        self._handles = []
        # Streams still alive when the interpreter exits crash `rocprofv3 --kernel-trace` in a static destructor after its tool
        # finalisation (profiles/r05/partition_kt.txt: SIGSEGV under __cxa_finalize; destroyed in order -> clean exit): a safety
        # net for callers that never call close() -- atexit handlers run before static destructors
        import atexit
        import weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())
        try:
            self.learner_stream = self._stream(learner)
            self.collection_streams = [self._stream(collect) for _ in range(int(collection_streams))]
        except Exception:
            self.close()                      # (no half-built partition: the streams created so far are destroyed)
            raise

    def _stream(self, mask):
        import torch
        arr = (C.c_uint32 * len(mask))(*mask)
        h = C.c_void_p()
        _lib.check(self._L.mrsim_stream_create_cu_mask(self.index, arr, len(mask), C.byref(h)), "mrsim_stream_create_cu_mask")
        self._handles.append(h)
        return torch.cuda.ExternalStream(h.value, device=self.device)

    @property
    def closed(self):
        return not self._handles

    def close(self):
        """Destroy the streams.  ORDER MATTERS: everything that still refers to them must be gone first -- the torch
        ExternalStream wrappers handed out by this object (a RolloutCollector built on `collection_streams` keeps them and the
        events it recorded on them), tensors allocated while one of them was the current stream (the caching allocator files a
        block under the stream it was allocated on) and graphs captured on them.  DDPG.close() does that for the training loop;
        a caller that used the streams directly drops its own references, then calls this.  The device is synchronised first,
        the allocator's cached blocks are returned, and only then are the streams destroyed; idempotent."""
        if not self._handles:
            return
        import gc
        import torch
        self.learner_stream, self.collection_streams = None, []
        gc.collect()                                   # wrappers / events that only cycles kept alive
        with torch.cuda.device(self.index):
            torch.cuda.synchronize()
            torch.cuda.empty_cache()                   # cached blocks filed under the streams about to go
            for h in self._handles:
                self._L.mrsim_stream_destroy(h)
        self._handles = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
