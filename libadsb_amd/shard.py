"""Multi-GPU use of the scanner: independent reference buffers are split over ranks, records come back to one rank.

Every 262144-byte reference buffer is demodulated independently, so a recording is partitioned at buffer
granularity with no halo and no exchange on the data path (rank r takes buffers [r*B/N, (r+1)*B/N), a trailing partial
buffer is never delivered: reference RTLSDR.hpp:419-442).  The only communication is the gather of the (small) record
arrays to the rank that runs the sequential resolver -- RCCL over xGMI with backend "nccl", gloo in the CPU tests:

  gather_records      every rank receives the whole stream (all_gather of counts, then of padded bytes)
  RootGather          only the root receives it: one fixed-size dist.gather per step, the record count travels in a
                      32-byte header in front of the records, so there is no separate count exchange; on "nccl" the payload
                      goes device to device (the scanner copies its sorted records straight into the send buffer)
  NodeGather          the same result on one node without funnelling the records through the root's GPU: every rank's GPU writes
                      its records over its own PCIe link into a page-locked segment of node-shared host memory; the 32-byte header
                      of a step (count, first buffer, rank, step) is written by the rank's host thread into its slot of the shared
                      control page once the copy's event has completed, and the root polls the slots -- no collective, no RCCL
                      kernel beside the persistent scan (round 4; until then the headers were gathered over RCCL).  With
                      RootGather the root's one host link carries N x 9 MB per GiB-step and bounds the job from N = 2 on; here it
                      carries 9 MB whatever N is.
"""
import mmap
import os

import numpy as np
import torch
import torch.distributed as dist

from . import RECORD_DTYPE

REC = RECORD_DTYPE.itemsize  # 32


def shard_range(nbuf_total, rank, world):
    """Contiguous block of buffers owned by `rank`: [first, first + count)."""
    first = rank * nbuf_total // world
    last = (rank + 1) * nbuf_total // world
    return first, last - first


def gather_records(records, first_buffer, group=None, device=None):
    """All ranks call this with their local records (buffer indices local to the shard) and the global index of the
    shard's first buffer.  Every rank receives the concatenation in rank (= stream) order with global buffer indices."""
    world = dist.get_world_size(group)
    device = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    rec = np.ascontiguousarray(records, dtype=RECORD_DTYPE).copy()
    rec["buffer"] += first_buffer
    count = torch.tensor([len(rec)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    payload = torch.zeros(cap * REC, dtype=torch.uint8)
    if len(rec):
        payload[:len(rec) * REC] = torch.from_numpy(rec.view(np.uint8).reshape(-1))
    payload = payload.to(device)
    parts = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(parts, payload, group=group)
    out = [np.frombuffer(p.cpu().numpy().tobytes()[:n * REC], dtype=RECORD_DTYPE) for p, n in zip(parts, counts)]
    return np.concatenate(out) if out else np.zeros(0, RECORD_DTYPE)


class RootGather:
    """Per-step gather of every rank's sorted records to rank `root`, for the sharded recorded-file job.

    Each rank owns a send buffer of 1 + cap records: record 0 is a header {count, first_buffer} and the scanner's records
    follow.  One dist.gather of that fixed size per step; the root then copies exactly count_r records of every part to
    page-locked host memory, rebases their buffer indices to the recording and hands the rank-ordered stream to the
    resolver.  `cap` is fixed for the job (the caller sizes it from a first scan, with slack); a step that exceeds it
    raises rather than truncating.
    """

    def __init__(self, cap_records, root=0, group=None, device=None, dtype=RECORD_DTYPE):
        assert np.dtype(dtype).itemsize == REC
        self.dtype = np.dtype(dtype)  # RECORD_DTYPE, or PACKED_DTYPE (same size, same first 18 bytes)
        self.group, self.root, self.cap = group, root, int(cap_records)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if self.on_device else torch.device("cpu"))
        nbytes = (1 + self.cap) * REC
        self.send = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.parts = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(self.world)] if self.rank == root else None
        self.host = torch.zeros(self.world * nbytes, dtype=torch.uint8).pin_memory() if (self.rank == root and self.on_device) else None
        # The header goes to the device by an asynchronous copy; a non-root rank does not wait for it, so the page-locked source of
        # step k must not be rewritten for step k + 1 before the copy has run: a small ring (a rank is never more than the scanner's
        # two slots ahead of its own side stream)
        self._hdrs = [torch.zeros(REC, dtype=torch.uint8).pin_memory() if self.on_device else torch.zeros(REC, dtype=torch.uint8) for _ in range(4)]
        self._calls = 0

    def records_ptr(self):
        """Device (or host) address where this rank's records of the step go: right behind the header."""
        return self.send.data_ptr() + REC

    def host_records_view(self):
        """gloo/CPU mode: numpy view of the record area of the send buffer (fill it, then call gather)."""
        return self.send.numpy()[REC:].view(self.dtype)

    def gather(self, count, first_buffer):
        """Collective.  `count` records of this rank already lie at records_ptr().  Returns on the root the concatenated
        RECORD_DTYPE array with recording-wide buffer indices (a view of internal page-locked memory, valid until the
        next call), elsewhere None."""
        if count > self.cap:
            raise RuntimeError("rank %d produced %d records, send buffer holds %d" % (self.rank, count, self.cap))
        src = self._hdrs[self._calls & 3]
        self._calls += 1
        hdr = src.numpy().view(np.uint64)
        hdr[0], hdr[1] = count, first_buffer
        self.send[:REC].copy_(src, non_blocking=True)
        dist.gather(self.send, self.parts, dst=self.root, group=self.group)
        if self.rank != self.root:
            return None
        # headers first (world x 32 bytes), then exactly the bytes that hold records
        heads = torch.stack([p[:REC] for p in self.parts]).cpu().numpy().view(np.uint64).reshape(self.world, -1)
        counts = [int(h[0]) for h in heads]
        firsts = [int(h[1]) for h in heads]
        total = sum(counts)
        if self.on_device:
            out = self.host[:total * REC]
            at = 0
            for p, n in zip(self.parts, counts):
                if n:
                    out[at:at + n * REC].copy_(p[REC:REC + n * REC], non_blocking=True)
                at += n * REC
            torch.cuda.current_stream().synchronize()
            rec = out.numpy().view(self.dtype)
        else:
            rec = np.concatenate([p.numpy()[REC:REC + n * REC].view(self.dtype) for p, n in zip(self.parts, counts)]) if total else np.zeros(0, self.dtype)
            rec = rec.copy()
        at = 0
        for n, f in zip(counts, firsts):
            if n and f:
                rec["buffer"][at:at + n] += f
            at += n
        return rec


class _PendingParts:
    """What NodeGather.gather(wait=False) returns on the root: result() waits for the step's headers (hence for every rank's
    copy) and returns the per-rank views."""

    def __init__(self, owner, step):
        self.owner, self.step = owner, step

    def result(self):
        return self.owner.collect(self.step)


class NodeGather:
    """Per-step hand-over of every rank's sorted records to rank `root` through node-shared page-locked host memory.

    One file in /dev/shm holds a control page and world x 2 page-aligned segments of cap records (two, so that the ranks can fill
    step k + 1 while the root still reads step k).  Every rank maps the file; a rank registers only ITS OWN two segments with the HIP
    runtime (they are the only ones its GPU writes: its scanner copies the step's records straight from HBM into `records_ptr(step)`
    over its own PCIe link), the root reads the others through the plain mapping.  `gather(step, count, first_buffer, event=...)`
    posts the step's 32-byte header (count, first buffer, rank, step): it is WRITTEN into the rank's slot of the control page by the
    rank's host thread once `event` -- recorded on the stream behind the copy -- has completed, the step number last and with release
    semantics (adsb_amd_shm_post_header: C11 atomics in libadsb_amd.so, not interpreter stores), so a header the root can see implies
    records it can see.  By default gather() does that before it returns (it waits for the event); with wait=False the header is only
    NOTED and goes out at the rank's next gather / acquire / flush / close -- the pipelined loop's form, where by then the copy is long
    done.  The root polls the
    slots (collect(), bounded) and gets one (records view, first buffer) pair per rank, in rank = recording order; the resolver
    consumes them one after the other without a copy (`concatenate` makes one array with recording-wide buffer indices where one is
    wanted).  Nothing per step goes through torch.distributed: the collective that used to carry the headers put one RCCL kernel per
    step beside the persistent scan kernel, about two of the four points the sharded step cost over the plain one.

    Flow control is explicit: a segment of parity k & 1 is rewritten for step k + 2, and nothing stops a rank from running that far
    ahead.  The root therefore publishes the last step it has finished READING in the control page (`release(step)`), and `acquire(step)` -- called by every
    rank before it lets its GPU write the segment of `step` -- waits until step - 2 has been released (bounded: raises TimeoutError).

    Raises OSError at construction, on every rank, when the segment cannot be set up on any one of them (no /dev/shm space,
    registration refused): callers fall back to RootGather.
    """

    PAGE = 4096
    _SLOT0, _SLOT = 16, 8  # control page as int64: [0] last step released; rank r's header of a step at [_SLOT0 + _SLOT * (2 r + (step & 1))]:
    #                        count, first buffer, rank, step (two per rank like the segments: a rank may be a step ahead of the root's reading)
    _fail_rank_for_tests = None  # (rank, stage): that rank fails at that stage of the construction ("file", "map", "register")

    def __init__(self, cap_records, root=0, group=None, tag=None, acquire_timeout_s=120.0, dtype=RECORD_DTYPE, device_writes=None):
        """device_writes: the ranks' GPUs write the segments (each rank registers its own two with the HIP runtime).  Default: whether the
        group's backend is "nccl".  True under gloo is the one-GPU-box rehearsal: N processes, every one's GPU context copying into its
        own registered pages of the shared file, the root reading all of them unregistered -- the production data path with gloo only
        for the set-up collectives."""
        assert np.dtype(dtype).itemsize == REC
        self.dtype = np.dtype(dtype)  # RECORD_DTYPE, or PACKED_DTYPE (same size, same first 18 bytes)
        self.group, self.root, self.cap = group, root, int(cap_records)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self.device_writes = self.on_device if device_writes is None else bool(device_writes)
        self.acquire_timeout_s = float(acquire_timeout_s)
        assert self._SLOT0 + self._SLOT * 2 * self.world <= self.PAGE // 8, "the control page holds %d ranks' headers" % ((self.PAGE // 8 - self._SLOT0) // self._SLOT // 2)
        self.seg = -(-self.cap * REC // self.PAGE) * self.PAGE  # whole pages: a rank registers its segments alone
        total = self.PAGE + self.world * 2 * self.seg
        dev = torch.device("cuda", torch.cuda.current_device()) if self.on_device else torch.device("cpu")
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        self._map = None
        self._np = None
        self._registered = []
        self._posted = []  # headers noted but not yet written: (step, count, first buffer, event)
        from . import lib as _native
        self._L = _native()  # the control page is only ever touched through its release / acquire helpers
        # The name carries a nonce chosen by the root for this instance (two jobs, or two instances of one job, never meet in one
        # file) and the file is created exclusively.
        nonce = torch.zeros(1, dtype=torch.int64, device=dev)
        if self.rank == root:
            nonce[0] = int.from_bytes(os.urandom(6), "little")
        dist.broadcast(nonce, src=root, group=group)
        self.path = "/dev/shm/libadsb_amd_gather_%s_%d_%012x" % (tag or os.environ.get("MASTER_PORT", "0"), os.getuid(), int(nonce.item()))

        def fails(stage):
            return self._fail_rank_for_tests == (self.rank, stage)
        try:
            if self.rank == root:
                if fails("file"):
                    raise OSError("forced failure (test)")
                fd = os.open(self.path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                try:
                    # The root sizes the file and allocates the control page; every rank allocates its own two segments below -- allocate,
                    # not just size (tmpfs answers a first write it has no room for with SIGBUS, fallocate with ENOSPC), and each rank its
                    # own so that the pages come from the memory of the node that rank runs on: with the root allocating everything, eight
                    # GPUs' hand-overs (8 x 33 GB/s) would all land in one socket's memory.
                    os.ftruncate(fd, total)
                    os.posix_fallocate(fd, 0, self.PAGE)
                finally:
                    os.close(fd)
        except OSError:
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)  # also the barrier after which the file exists
        if int(ok.item()):
            try:
                if fails("map"):
                    raise OSError("forced failure (test)")
                fd = os.open(self.path, os.O_RDWR)
                try:
                    os.posix_fallocate(fd, self._offset(self.rank, 0), 2 * self.seg)  # this rank's two segments are adjacent
                    self._map = mmap.mmap(fd, total)
                finally:
                    os.close(fd)
                self._np = np.frombuffer(self._map, dtype=np.uint8)
                self._ctl_addr = self._np.ctypes.data  # the control page as int64 words: [0] last step the root has finished reading, -1 before any
                if self.rank == root:
                    self._L.adsb_amd_shm_store_release(self._word(0), -1)
                for par in (0, 1):
                    self._L.adsb_amd_shm_post_header(self._word(self._slot(self.rank, par)), 0, 0, self.rank, -1)  # no step posted yet
                for step in (0, 1):  # first touch of this rank's own segments (on the NUMA node this rank runs on)
                    o = self._offset(self.rank, step)
                    self._np[o:o + self.seg] = 0
                if self.device_writes:
                    if fails("register"):
                        raise OSError("forced failure (test)")
                    for step in (0, 1):
                        ptr = self._np.ctypes.data + self._offset(self.rank, step)
                        rc = torch.cuda.cudart().cudaHostRegister(ptr, self.seg, 0)
                        if int(rc) != 0:
                            raise OSError("hipHostRegister failed (%s)" % (rc,))
                        self._registered.append(ptr)
            except (OSError, ValueError, RuntimeError):
                ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if self.rank == root and os.path.exists(self.path):
            os.unlink(self.path)  # every rank holds its mapping; the name is not needed any more
        if not int(ok.item()):
            self.close()
            raise OSError("node-shared record segment could not be set up on every rank")
        self.ranks_seen = 0

    def _slot(self, rank, step):
        return self._SLOT0 + self._SLOT * (2 * rank + (step & 1))

    def _offset(self, rank, step):
        return self.PAGE + (rank * 2 + (step & 1)) * self.seg

    def _word(self, index):
        """Address of int64 word `index` of the control page."""
        return self._ctl_addr + 8 * index

    def _released(self):
        return int(self._L.adsb_amd_shm_load_acquire(self._word(0)))

    def flush(self):
        """Write the headers noted so far into this rank's slot of the control page, oldest first, each once its event has completed
        (waits for it: by the time this is called -- the rank's next step, or the end of a loop -- the copy is long done)."""
        while self._posted:
            step, count, first, event = self._posted[0]  # (taken off the list only once its header is written: a wait or a write that raises
            if event is not None:                         # leaves it, and everything behind it, noted for the next flush or close)
                event.synchronize()
            self._L.adsb_amd_shm_post_header(self._word(self._slot(self.rank, step)), count, first, self.rank, step)
            self._posted.pop(0)

    def acquire(self, step):
        """Before this rank's segment of `step` is written: wait until the root has finished reading step - 2 (same segment)."""
        self.flush()  # the header of the step before, if its copy has an event: the root may be waiting for it before it releases
        need = step - 2
        if need < 0 or self._released() >= need:
            return
        import time
        deadline = time.monotonic() + self.acquire_timeout_s
        while self._released() < need:
            if time.monotonic() > deadline:
                raise TimeoutError("rank %d: the root has not released step %d after %.0f s (last released: %d)"
                                   % (self.rank, need, self.acquire_timeout_s, self._released()))
            time.sleep(0)

    def release(self, step):
        """Root: the views of `step` (and of every earlier step) will not be read any more."""
        if self.rank == self.root and self._released() < step:
            self._L.adsb_amd_shm_store_release(self._word(0), int(step))

    def records_ptr(self, step):
        """Host address (page-locked, device-writable) where this rank's records of `step` go.  Call acquire(step) first."""
        return self._np.ctypes.data + self._offset(self.rank, step)

    def host_records_view(self, step):
        """numpy view of this rank's segment for `step` (gloo/CPU mode: acquire(step), fill it, then call gather)."""
        o = self._offset(self.rank, step)
        return self._np[o:o + self.cap * REC].view(self.dtype)

    def gather(self, step, count, first_buffer, wait=True, event=None):
        """`count` records of this rank lie (or, with `event`, will lie once it has completed) in its segment of `step`.  Not a
        collective.  Returns on the root [(records, first_buffer)] per rank (views of the shared segments, valid until
        release(step)), elsewhere None.
        wait=True (default): self-completing -- the call waits for the event (on "nccl" none given means: one recorded here, on the
        current stream) and writes this rank's header before it returns; the root then waits for every rank's header.
        wait=False: the pipelined form.  The header is only noted; it is written when the rank comes back (its next gather / acquire,
        flush(), or close()), by when the copy is long done, and the root gets a handle whose result() gives the views once every
        rank's header of the step is there -- so that it can enqueue step k + 1 before it looks at step k, like the other ranks do."""
        if count > self.cap:
            raise RuntimeError("rank %d produced %d records, its segment holds %d" % (self.rank, count, self.cap))
        if event is None and self.device_writes:
            event = torch.cuda.current_stream().record_event()  # the records were copied on the current stream (the documented use)
        self._posted.append((int(step), int(count), int(first_buffer), event))
        if wait or event is None:
            self.flush()
        if self.rank != self.root:
            return None
        pending = _PendingParts(self, step)
        return pending.result() if wait else pending

    def collect(self, step):
        """Root: wait (bounded) until every rank's header of `step` is in the control page; the per-rank views."""
        import time
        self.flush()
        deadline = time.monotonic() + self.acquire_timeout_s
        out, r = [], 0
        hdr = np.zeros(4, dtype=np.int64)
        while r < self.world:
            base = self._word(self._slot(r, step))
            if not self._L.adsb_amd_shm_read_header(base, int(step), hdr.ctypes.data):
                if time.monotonic() > deadline:
                    self._L.adsb_amd_shm_read_header(base, -(1 << 62), hdr.ctypes.data)
                    raise TimeoutError("root: no header of rank %d for step %d after %.0f s (its last: step %d)" % (r, step, self.acquire_timeout_s, int(hdr[3])))
                time.sleep(0)
                continue
            n, first, who, st = (int(x) for x in hdr)
            if st != step or who != r:
                raise RuntimeError("header of rank %d for step %d carries step %d, rank %d" % (r, step, st, who))
            o = self._offset(r, step)
            out.append((self._np[o:o + n * REC].view(self.dtype), first))
            r += 1
        self.ranks_seen = len(out)
        return out

    @staticmethod
    def concatenate(parts):
        """One array with recording-wide buffer indices from what gather returned."""
        recs = []
        for rec, first in parts:
            rec = rec.view(np.uint8).copy().view(rec.dtype)  # a byte copy: numpy copies structured arrays field by field, 6 ms for 9 MB
            if first:
                rec["buffer"] += first
            recs.append(rec)
        return np.concatenate(recs) if recs else np.zeros(0, RECORD_DTYPE)

    def close(self):
        """Also writes out any header that was only noted (a rank's last step in the pipelined form): a rank that is done does not
        leave the root waiting for it."""
        if self._np is not None and self._posted:
            try:
                self.flush()
            except Exception:  # a failed device at teardown must not mask the error that brought us here
                self._posted = []
        for ptr in self._registered:
            torch.cuda.cudart().cudaHostUnregister(ptr)
        self._registered = []
        self._np = None
        self._ctl_addr = None
        if self._map is not None:
            try:
                self._map.close()
            except BufferError:
                pass  # views handed out are still alive; the mapping goes with the process
            self._map = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Watchdog:
    """A rank that waits longer than `limit_s` in one phase prints where it is (rank, step, phase) and ends its process with a
    non-zero code: a stuck collective must end as a failed process, never as a hang.  phase(name, step) marks progress."""

    def __init__(self, rank, limit_s=300.0):
        import threading
        import time
        self.rank, self.limit_s = rank, float(limit_s)
        self._state = ("start", -1, time.monotonic())
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="adsb-watchdog", daemon=True)
        self._thread.start()

    def phase(self, name, step=-1):
        import time
        self._state = (name, step, time.monotonic())

    def _run(self):
        import sys
        import time
        while not self._stop.wait(min(1.0, self.limit_s / 4)):
            name, step, since = self._state
            if time.monotonic() - since > self.limit_s:
                sys.stderr.write("watchdog: rank %d has been in phase '%s' of step %d for more than %.0f s -- giving up\n" % (self.rank, name, step, self.limit_s))
                sys.stderr.flush()
                os._exit(3)

    def stop(self):
        self._stop.set()


# ---- UAT 978: one process_buffer over a stream that is cut over several GPUs (adsb_amd_uat_part_scan / _finish) -----------------------
UAT_FRAME_SAMPLES = 2 * (36 + 4416)       # the longest frame, sync word included
UAT_LEAD_SAMPLES = 64                     # before a part's own samples: the loop can fire at a start bit 17 bits before the bit it stands at
UAT_TAIL_SAMPLES = 3 * UAT_FRAME_SAMPLES + 64  # behind them: a frame that starts in the part, a frame behind its stale-register window, and slack


def uat_part(nsamples, rank, world, align=64):
    """The part of rank `rank` when a stream of `nsamples` is cut over `world`: (window first sample, window end, own begin, own end, last).
    Own ranges are contiguous multiples of `align` samples; a part whose tail would reach past the stream's end takes the rest of the
    stream and is the last one (later ranks get an empty part: own begin == own end)."""
    per = -(-nsamples // world)
    per = -(-per // align) * align
    begin = min(rank * per, nsamples)
    end = min(begin + per, nsamples)
    # the first part whose tail does not fit closes the stream
    closing = next((r for r in range(world) if min((r + 1) * per, nsamples) + UAT_TAIL_SAMPLES >= nsamples), world - 1)
    if rank > closing:
        return nsamples, nsamples, nsamples, nsamples, False
    last = rank == closing
    if last:
        end = nsamples
    w0 = max(begin - UAT_LEAD_SAMPLES, 0)
    w1 = nsamples if last else end + UAT_TAIL_SAMPLES
    return w0, w1, begin, end, last


def uat_run_parts(handles, device_ptr, nsamples, offset=0, collect=True):
    """The parts one after the other on the handles given (one per part; they may sit on one GPU or on several, `device_ptr` being the
    stream's address on each): (frames in stream order, consumed) as Uat978.process_device gives them for the whole stream.  The heavy
    half (part_scan) of every part is issued first; only the 8-byte "where the loop stands" travels from part to part."""
    world = len(handles)
    parts = [uat_part(nsamples, r, world) for r in range(world)]
    for u, (w0, w1, b, e, last) in zip(handles, parts):
        if e > b:
            u.part_scan(device_ptr + 2 * w0, w1 - w0)
    frames, at, consumed = [], 0, None
    for u, (w0, w1, b, e, last) in zip(handles, parts):
        if e <= b:
            continue
        fr, exit_local, done = u.part_finish(b - w0, e - w0, max(at - w0 // 2, 0), last, offset=offset + w0, collect=collect)
        frames += fr
        at = exit_local + w0 // 2
        if last:
            consumed = done + w0
    return frames, consumed


def uat_chain_step(u, dist, word, rank, world, part, window_ptr, collect=False, phase=None):
    """One rank's share of ONE process_buffer over a stream cut over the ranks (uat_part): the heavy half of its part at once (part_scan),
    then the 8-byte "where the loop stands" from the rank before (recv), the decisions and up-calls of its own part (part_finish), and the
    loop's position on to the rank behind (send).  `word`: a one-element int64 tensor on the device the process group's backend moves (cuda for
    nccl, cpu for gloo).  A rank whose part is empty (the stream closed before it) passes the word on unchanged.  A failure travels down the
    chain as -1, so that no later rank sits in recv until a watchdog fires.  Returns (frames, samples consumed counted from the stream's start --
    meaningful on the closing rank --, the loop's position after this part)."""
    w0, w1, b, e, last = part
    mark = phase or (lambda *_: None)
    mark("part scan")
    if e > b:
        u.part_scan(window_ptr, w1 - w0)
    at = 0
    if rank > 0:
        mark("waiting for the rank before")
        dist.recv(word, src=rank - 1)
        at = int(word.item())
        if at < 0:
            if rank < world - 1:
                dist.send(word, dst=rank + 1)
            raise RuntimeError("rank %d: the part of an earlier rank failed" % rank)
    mark("part finish")
    frames, done, exit_bit = [], 0, at
    if e > b:
        try:
            frames, exit_local, done = u.part_finish(b - w0, e - w0, max(at - w0 // 2, 0), last, offset=w0, collect=collect)
        except Exception:
            if rank < world - 1:
                word[0] = -1
                dist.send(word, dst=rank + 1)
            raise
        exit_bit = exit_local + w0 // 2
    if rank < world - 1:
        word[0] = exit_bit
        dist.send(word, dst=rank + 1)
    return frames, done + w0, exit_bit
