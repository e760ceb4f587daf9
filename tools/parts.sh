#!/bin/bash
# tools/parts.sh : cost of scan1090_kernel by part -- builds with the later parts compiled out (ab_ship/part1..part4.so, made by
#   for k in 1 2 3; do EXTRA_FLAGS="-DADSB_AMD_DIAG_BUILD=1 -DDIAG_PARTS=$k" tools/build_variant.sh WORK part$k; done; tools/build_variant.sh WORK part4
# on the build host), PMC instruction counts and kernel time of each on the 1 GiB bench workload.
tools/pmc_ab.sh part1 part2 part3 part4 > gpurun_out/parts_pmc.txt 2>&1
python3 tools/ab.py lib=ab_ship/part1.so lib=ab_ship/part2.so lib=ab_ship/part3.so lib=ab_ship/part4.so > gpurun_out/parts_time.txt 2>&1
python3 - <<'PY'
import re
pmc = open("gpurun_out/parts_pmc.txt").read()
times = dict(re.findall(r"part(\d)\.so'\}\s+kernel_ms median ([0-9.]+)", open("gpurun_out/parts_time.txt").read()))
rows = {}
for name, body in re.findall(r"== (part\d)\n((?:   .*\n)+)", pmc):
    rows[name] = {m[0]: float(m[1]) for m in re.findall(r"(\w+)\s+mean [0-9.e+]+\s+per chunk ([0-9.]+)", body)}
print("scan1090_kernel by part, per 4096-position chunk (PMC, 1 GiB = 131072 chunks); kernel time = median of the in-process A/B")
print("%-34s %8s %8s %8s %14s %10s" % ("build", "VALU", "SALU", "LDS", "VALU busy qcyc", "kernel ms"))
names = {"part1": "window load + s", "part2": "+ stage 1", "part3": "+ survivor queue + stage 2", "part4": "+ demodulation (complete)"}
prev = None
for k in ("part1", "part2", "part3", "part4"):
    r = rows.get(k, {})
    print("%-34s %8.0f %8.0f %8.0f %14.0f %10s" % (names[k], r.get("SQ_INSTS_VALU", 0), r.get("SQ_INSTS_SALU", 0), r.get("SQ_INSTS_LDS", 0), r.get("SQ_ACTIVE_INST_VALU", 0), times.get(k[-1], "?")))
    if prev:
        print("%-34s %+8.0f %+8.0f %+8.0f %+14.0f" % ("   of which this part", r.get("SQ_INSTS_VALU", 0) - prev.get("SQ_INSTS_VALU", 0), r.get("SQ_INSTS_SALU", 0) - prev.get("SQ_INSTS_SALU", 0),
                                                     r.get("SQ_INSTS_LDS", 0) - prev.get("SQ_INSTS_LDS", 0), r.get("SQ_ACTIVE_INST_VALU", 0) - prev.get("SQ_ACTIVE_INST_VALU", 0)))
    prev = r
PY
