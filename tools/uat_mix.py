"""Kernel time of the UAT path by frame mix: 256 MiB synthetic streams with parts of the default mix switched off.
    python tools/uat_mix.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

piece, npieces = 64 << 20, 4
u = A.Uat978(0)
for name, over in (("default mix", {}), ("no uplink", dict(pct_uplink=0)), ("no uplink, nothing corrupted", dict(pct_uplink=0, pct_corrupt=0)),
                   ("long clean frames only", dict(pct_uplink=0, pct_corrupt=0, pct_long=100)),
                   ("short clean frames only", dict(pct_uplink=0, pct_corrupt=0, pct_long=0)),
                   ("uplink only", dict(pct_uplink=100, pct_corrupt=0)), ("noise only", dict(mean_gap_bits=1 << 30))):
    cfg = synth.default_cfg978(**over)
    dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
    with ThreadPoolExecutor(npieces) as ex:
        for k, h in enumerate(ex.map(lambda k: synth.fill978(k, piece, cfg), range(npieces))):
            dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
    torch.cuda.synchronize()
    frames, _ = u.process_device(dev.data_ptr(), dev.numel() // 2)
    best = None
    for _ in range(4):
        c0 = u.timing()["candidates"]
        u.process_device(dev.data_ptr(), dev.numel() // 2, collect=False)
        tm = u.timing()
        tm["matches"] = tm["candidates"] - c0
        hw = tm["host_wall_ms"]
        if best is None or tm["demod_ms"] < best[0]["demod_ms"]:
            best = (tm, hw)
    print("%-32s frames %6d matches %6d  scan %.3f ms  demod %.3f ms  host %s" % (name, len(frames), best[0]["matches"], best[0]["scan_ms"],
                                                                                best[0]["demod_ms"], {k: round(v, 3) for k, v in best[1].items()}), flush=True)
