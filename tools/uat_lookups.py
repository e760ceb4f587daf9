import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import libadsb_amd as A
from libadsb_amd import synth
from concurrent.futures import ThreadPoolExecutor
piece, npieces = 64 << 20, 16
cfg = synth.default_cfg978()
dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
with ThreadPoolExecutor(16) as ex:
    for k, h in enumerate(ex.map(lambda k: synth.fill978(k, piece, cfg), range(npieces))):
        dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
torch.cuda.synchronize()
u = A.Uat978(0)
u.process_device(dev.data_ptr(), dev.numel() // 2, collect=False)
e0 = u.timing()["extra_lookups"]
for _ in range(3):
    u.process_device(dev.data_ptr(), dev.numel() // 2, collect=False)
    t = u.timing()
    print("extra lookups per call", t["extra_lookups"] - e0, "loop ms", t["host_wall_ms"]["loop"])
    e0 = t["extra_lookups"]
