"""N > 1 path on CPU: buffer sharding + record gather over gloo (world size 2), resolver on the gathered stream."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np
import torch.distributed as dist
import libadsb_amd as A
from libadsb_amd import synth, shard
import helpers as H
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
NBUF, BB = 9, A.REF_BUFFER_BYTES
first, count = shard.shard_range(NBUF, rank, world)
iq, _ = synth.fill_range(first, count)
# the GPU scanner's output contract, produced here from the oracle's probes (no GPU in this test)
local = H.expected_records(iq, BB)
allrec = shard.gather_records(local, first)
if rank == 0:
    full, _ = synth.fill_range(0, NBUF)
    H.assert_records_equal(allrec, H.expected_records(full, BB))
    n, fr, ac = A.Resolver().feed(allrec, BB // 2, NBUF)
    ofr, oac = H.oracle_run(full, BB)
    H.assert_streams_equal(fr, ac, ofr, oac)
    print('OK', n, len(allrec))
# the per-step form bench.py uses: fixed-size gather to the root only, count in the header
rg = shard.RootGather(1000)  # the same capacity on every rank (bench.py agrees on it with an all_reduce MAX)
for step in range(2):
    rg.host_records_view()[:len(local)] = local
    got = rg.gather(len(local), first)
    if rank == 0:
        H.assert_records_equal(got, allrec)
    else:
        assert got is None
try:
    shard.RootGather(3).gather(4, 0)
    raise SystemExit('an over-full send buffer must raise')
except RuntimeError:
    pass
if rank == 0:
    print('ROOT-GATHER-OK')
# the node-local form: every rank fills its own segment of shared memory, only headers are gathered
ng = shard.NodeGather(1000)
import time
for step in range(7):
    keep = len(local) - (step %% 3)  # a different count per step: stale records of step - 2 must not leak
    ng.acquire(step)
    ng.host_records_view(step)[:keep] = local[:keep]
    parts = ng.gather(step, keep, first)
    if rank == 0:
        assert ng.ranks_seen == world
        assert [f for _, f in parts] == [shard.shard_range(NBUF, r, world)[0] for r in range(world)]
        if step == 3:
            time.sleep(0.5)  # a slow consumer: the other rank reaches acquire(5) long before step 3 is released and must wait there
        got = shard.NodeGather.concatenate(parts)
        assert len(parts[0][0]) == keep
        mine = np.array(local[:keep]); H.assert_records_equal(got[:keep], mine)
        if step %% 3 == 0:
            H.assert_records_equal(got, allrec)
        # the resolver reads the per-rank parts in place, one after the other: same stream as from the concatenated array
        if step == 0:
            r1, r2 = A.Resolver(), A.Resolver()
            acc = 0
            for (part, f0), r in zip(parts, range(world)):
                n_, fr_, ac_ = r1.feed(part, BB // 2, shard.shard_range(NBUF, r, world)[1])
                acc += n_
            n2, _, _ = r2.feed(got, BB // 2, NBUF)
            assert acc == n2 and r1.aircraft_count() == r2.aircraft_count()
        ng.release(step)
    else:
        assert parts is None
# a header noted with an event is written only when the rank comes back, and only after the event has been waited for
class Ev:
    waited = False
    def synchronize(self):
        self.waited = True
lazy = shard.NodeGather(16, tag='lazy', acquire_timeout_s=0.3)
ev = Ev()
lazy.acquire(0)
lazy.host_records_view(0)[:2] = local[:2]
def posted_step(g, step):  # the step number in this rank's header slot of the control page
    return int(g._L.adsb_amd_shm_load_acquire(g._word(g._slot(rank, step) + 3)))
h = lazy.gather(0, 2, first, wait=False, event=ev)
assert not ev.waited and posted_step(lazy, 0) == -1
if rank == 0:
    try:
        if world > 1:
            h.result()  # rank 0's own header gets written by its flush, the others' have not been: bounded wait, then an error naming the rank
            raise SystemExit('collect must not see headers that were never written')
    except TimeoutError as e:
        assert 'rank 1' in str(e)
dist.barrier()
lazy.flush()
assert ev.waited and posted_step(lazy, 0) == 0
dist.barrier()
if rank == 0:
    parts = h.result()
    assert lazy.ranks_seen == world and all(len(p) == 2 for p, _ in parts)
    lazy.release(0)
# the default form is self-completing: gather() waits for the event and posts the header before it returns (no "must come back" rule)
ev1 = Ev()
lazy.acquire(1)
lazy.host_records_view(1)[:1] = local[:1]
parts = lazy.gather(1, 1, first, event=ev1)
assert ev1.waited and posted_step(lazy, 1) == 1
assert (parts is not None) == (rank == 0)
if rank == 0:
    assert all(len(p) == 1 for p, _ in parts)
    lazy.release(1)
# and a header that was only noted goes out when the rank closes its side (a rank's last step in the pipelined form)
dist.barrier()
ev2 = Ev()
lazy.acquire(2)
h2 = lazy.gather(2, 0, first, wait=False, event=ev2)
if rank != 0:
    lazy.close()
    assert ev2.waited
dist.barrier()
if rank == 0:
    assert len(h2.result()) == world
    lazy.close()
# flow control: without a release the writer of step + 2 must not get its segment
slow = shard.NodeGather(8, tag='credit', acquire_timeout_s=0.3)
for step in range(2):
    slow.acquire(step)
    slow.gather(step, 0, first)
try:
    slow.acquire(2)
    raise SystemExit('acquire(2) must wait for release(0)')
except TimeoutError:
    pass
dist.barrier()
slow.release(0)
dist.barrier()
slow.acquire(2)
try:
    shard.NodeGather(3, tag='small').gather(0, 4, 0)
    raise SystemExit('an over-full segment must raise')
except RuntimeError:
    pass
# a rank that cannot set its segment up (here: rank 1, at every stage in turn) takes every rank to the fallback
for stage in ('file', 'map', 'register'):
    shard.NodeGather._fail_rank_for_tests = (0 if stage == 'file' else 1, stage)
    try:
        shard.NodeGather(16, tag='fail-' + stage)
        ok = stage == 'register'  # gloo: nothing is registered, so that stage cannot fail here
    except OSError:
        ok = stage != 'register'
    shard.NodeGather._fail_rank_for_tests = None
    assert ok, stage
    fb = shard.RootGather(16)  # and the fallback works right after
    fb.host_records_view()[:3] = local[:3]
    got = fb.gather(3, first)
    assert (got is not None) == (rank == 0)
import glob
assert not glob.glob('/dev/shm/libadsb_amd_gather_*fail-*'), 'a failed construction must not leave its file behind'
if rank == 0:
    print('NODE-GATHER-OK')
dist.barrier()
dist.destroy_process_group()
"""


def test_shard_ranges_cover_everything():
    from libadsb_amd import shard
    for n in (1, 7, 8, 4096, 32768):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1


@pytest.mark.parametrize("world,port", [(2, 29517), (5, 29518)])
def test_two_rank_gather_and_resolve(tmp_path, native_libs, world, port):
    """world 2, and world 5 (nine buffers over five ranks: uneven shards, four non-root ranks writing their own segments)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK" in out.stdout and "ROOT-GATHER-OK" in out.stdout and "NODE-GATHER-OK" in out.stdout


# BASELINE configs[3] is 32 768 buffers over 8 GPUs.  Its partition arithmetic, and the hand-over with eight ranks on a small recording:
# uneven shards (11 buffers), ranks with nothing to scan (5 buffers), the pipelined (noted-header) form the bench uses, and a rank that
# cannot set its segment up at each construction stage -> every rank falls back to the record gather together.
WORKER8 = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np
import torch.distributed as dist
import libadsb_amd as A
from libadsb_amd import synth, shard
import helpers as H
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
BB = A.REF_BUFFER_BYTES
assert shard.shard_range(32768, rank, 8) == (4096 * rank, 4096)  # the configuration itself: 1 GiB per GPU, no remainder
for NBUF in (11, 5):
    first, count = shard.shard_range(NBUF, rank, world)
    assert count in ((1, 2) if NBUF == 11 else (0, 1))
    iq = synth.fill_range(first, count)[0] if count else np.zeros(0, np.uint8)
    local = H.expected_records(iq, BB) if count else np.zeros(0, A.RECORD_DTYPE)
    ng = shard.NodeGather(2000, tag='w8-%%d' %% NBUF)
    full = synth.fill_range(0, NBUF)[0] if rank == 0 else None
    want = H.expected_records(full, BB) if rank == 0 else None
    prev = None
    for step in range(5):  # the bench's loop: note the header, come back a step later
        ng.acquire(step)
        ng.host_records_view(step)[:len(local)] = local
        h = ng.gather(step, len(local), first, wait=False)
        if prev is not None and rank == 0:
            parts = prev.result()
            assert ng.ranks_seen == 8 and [f for _, f in parts] == [shard.shard_range(NBUF, r, 8)[0] for r in range(8)]
            H.assert_records_equal(shard.NodeGather.concatenate(parts), want)
            ng.release(prev.step)
        prev = h
    if rank == 0:
        parts = prev.result()
        got = shard.NodeGather.concatenate(parts)
        H.assert_records_equal(got, want)
        n, fr, ac = A.Resolver().feed(got, BB // 2, NBUF)
        ofr, oac = H.oracle_run(full, BB)
        H.assert_streams_equal(fr, ac, ofr, oac)
        ng.release(prev.step)
    ng.close()
    for stage, who in (('file', 0), ('map', 5), ('register', 7)):
        shard.NodeGather._fail_rank_for_tests = (who, stage)
        try:
            shard.NodeGather(16, tag='w8-fail-' + stage)
            ok = stage == 'register'  # gloo: nothing is registered, so that stage cannot fail here
        except OSError:
            ok = stage != 'register'
        shard.NodeGather._fail_rank_for_tests = None
        assert ok, stage
        fb = shard.RootGather(2000)  # all eight ranks build the fallback together and it delivers the same stream
        fb.host_records_view()[:len(local)] = local
        got = fb.gather(len(local), first)
        if rank == 0:
            H.assert_records_equal(got, want)
        else:
            assert got is None
# ---- the UAT workload's partition of configs[3]'s kind of job (bench.py --workload uat978 --gpus 8: one_stream_cut_over_ranks): ONE stream cut
# over eight ranks (shard.uat_part) and the chain that carries the loop's position from rank to rank (shard.uat_chain_step, the very function
# bench.py runs), with a stand-in for the GPU handle that records what it is asked: the product has no CPU path, so what is rehearsed here is the
# arithmetic of the cut, the order of the chain, the pass-through of a rank without a part and the way a failure travels down the chain.
import torch
LEAD, TAIL = shard.UAT_LEAD_SAMPLES, shard.UAT_TAIL_SAMPLES
for NS in (8 * (1 << 29), 8 * (1 << 29) + 12345, 5 * TAIL + 999, 2 * TAIL + 64, 1000):  # 8 x 1 GiB of u8 IQ; a ragged one; streams too short for eight parts
    parts = [shard.uat_part(NS, r, world) for r in range(world)]
    own = [(b, e) for _, _, b, e, _ in parts if e > b]
    assert own[0][0] == 0 and own[-1][1] == NS and all(own[k][1] == own[k + 1][0] for k in range(len(own) - 1)), (NS, own)  # the own ranges tile the stream
    assert sum(1 for p in parts if p[4]) == 1 and all(p[2] == p[3] == NS for p in parts[[p[4] for p in parts].index(True) + 1:])  # one closing part, nothing behind it
    for w0, w1, b, e, last in parts:
        if e > b:  # what adsb_amd_uat_part_finish asks of a window
            assert b %% 64 == 0 and (b == 0 or b - w0 >= 64) and (last and w1 == NS or not last and w1 - e >= TAIL) and 0 <= w0 <= b <= e <= w1 <= NS
    if NS == 8 * (1 << 29):
        assert [(b, e) for _, _, b, e, _ in parts] == [(r << 29, (r + 1) << 29) for r in range(8)]  # 1 GiB of IQ per GPU, no remainder

    class Stand:
        # exit position = own end in bits + 7 * (rank + 1): behind the part, different on every rank, so a word taken from the wrong rank shows
        def __init__(self, fail=False):
            self.calls, self.fail = [], fail
        def part_scan(self, ptr, n):
            self.calls.append(('scan', ptr, n))
        def part_finish(self, own_begin, own_end, entry, last, offset=0, collect=True):
            self.calls.append(('finish', own_begin, own_end, entry, last, offset))
            if self.fail:
                raise RuntimeError('stand-in failure')
            return [(rank, entry + offset // 2)], (own_end + offset) // 2 + 7 * (rank + 1) - offset // 2, own_end
    word = torch.zeros(1, dtype=torch.int64)
    u = Stand()
    w0, w1, b, e, last = parts[rank]
    frames, consumed, exit_bit = shard.uat_chain_step(u, dist, word, rank, world, parts[rank], 4096 + 2 * w0)
    exits = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(exits, torch.tensor([exit_bit], dtype=torch.int64))
    exits = [int(x.item()) for x in exits]
    before = 0 if rank == 0 else exits[rank - 1]
    if e > b:
        assert u.calls[0] == ('scan', 4096 + 2 * w0, w1 - w0) and u.calls[1] == ('finish', b - w0, e - w0, max(before - w0 // 2, 0), last, w0), u.calls
        assert frames == [(rank, max(before - w0 // 2, 0) + w0 // 2)] and exit_bit == e // 2 + 7 * (rank + 1) and consumed == e
    else:
        assert u.calls == [] and frames == [] and exit_bit == before  # no part: the word goes through unchanged
    dist.barrier()
    # a rank whose part fails: every rank behind it raises at once (-1 down the chain), the ranks before it are not disturbed
    nparts = len(own)
    bad = nparts // 2
    u = Stand(fail=rank == bad)
    try:
        shard.uat_chain_step(u, dist, word, rank, world, parts[rank], 4096 + 2 * w0)
        raised = False
    except RuntimeError as ex:
        raised = True
        assert ('stand-in' in str(ex)) == (rank == bad)
    assert raised == (rank >= bad), (NS, rank, bad, raised)
    dist.barrier()
import glob
assert not glob.glob('/dev/shm/libadsb_amd_gather_*w8-*'), 'no file may be left behind'
if rank == 0:
    print('EIGHT-RANK-OK')
dist.barrier()
dist.destroy_process_group()
"""


def test_eight_rank_rehearsal_of_the_recorded_file_partition(tmp_path, native_libs):
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "EIGHT-RANK-OK" in out.stdout


def test_bench_started_bare_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no RANK in the environment (how the driver starts it) must start the N ranks as child
    processes of torch.distributed.run on 127.0.0.1 and hand their return code back; with RANK set it must not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
        raise AssertionError("main() must exit with the children's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
