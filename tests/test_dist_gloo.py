"""N > 1 path on CPU: buffer sharding + record gather over gloo (world size 2), resolver on the gathered stream."""
import os
import subprocess
import sys

import numpy as np

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


def test_two_rank_gather_and_resolve(tmp_path, native_libs):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK" in out.stdout
