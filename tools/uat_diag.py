"""Where the demodulating wave's time goes: shader-clock cycles by phase and kind of match, from a measurement build (-DADSB_AMD_DIAG_BUILD=1 -DDIAG_UAT=1, diag.hip.h).
    EXTRA_FLAGS="-DADSB_AMD_DIAG_BUILD=1 -DDIAG_UAT=1" tools/build_variant.sh WORK uat_diag && ADSB_AMD_LIB=ab_ship/uat_diag.so python tools/uat_diag.py [MiB]
Cycles are per wave (one wave per position), summed over the launch; the table gives the mean per position and the share of the total.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import libadsb_amd as A  # noqa: E402
from libadsb_amd import synth  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
piece = 64 << 20
npieces = max(1, (mib << 20) // piece)
cfg = synth.default_cfg978()
dev = torch.empty(npieces * piece, dtype=torch.uint8, device="cuda")
with ThreadPoolExecutor(min(16, npieces)) as ex:
    for k, h in enumerate(ex.map(lambda k: synth.fill978(k, piece, cfg), range(npieces))):
        dev[k * piece:(k + 1) * piece].copy_(torch.from_numpy(h))
torch.cuda.synchronize()
u = A.Uat978(0)
L = A.lib()
L.adsb_amd_uat_diag.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
u.process_device(dev.data_ptr(), dev.numel() // 2, collect=False)
L.adsb_amd_uat_diag(out)  # reset
u.process_device(dev.data_ptr(), dev.numel() // 2, collect=False)
tm = u.timing()
assert L.adsb_amd_uat_diag(out) == 0
names = ["stage tile 0", "sync re-check", "slice", "syndromes", "RS decode", "further tiles", "output + chase"]
total = sum(out[k * 8 + p] for k in range(2) for p in range(7))
print("demod kernel %.3f ms; %d positions; %.3e wave-cycles in all" % (tm["demod_ms"], out[7] + out[15], total))
for k, kind in enumerate(("ADS-B", "uplink")):
    n = max(1, out[k * 8 + 7])
    print("%s: %d positions, %.0f cycles each, %.1f %% of all wave-cycles" % (kind, n, sum(out[k * 8 + p] for p in range(7)) / n,
                                                                             100.0 * sum(out[k * 8 + p] for p in range(7)) / total))
    for p in range(7):
        print("   %-16s %9.0f cycles per position  %5.1f %%" % (names[p], out[k * 8 + p] / n, 100.0 * out[k * 8 + p] / total))
