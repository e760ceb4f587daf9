"""profiles/traffic.json from the tracked rocprofv3 summaries: HBM bytes per launch of the dominant kernel of each workload.

    python tools/traffic_from_profile.py

Reads the newest of profiles/r0N_final_rocprof_summary.txt (1090: scan1090_kernel), profiles/r0N_uat978_rocprof_summary.txt and
profiles/r0N_mode2400_rocprof_summary.txt that holds both counters for its kernel.  FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KB per
dispatch (separate --pmc passes, tools/prof.sh / tools/uat_pmc.sh); on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, section HBM), so traffic = 2 x FETCH_SIZE + WRITE_SIZE.  bench.py reports the figure as roofline.traffic."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def means(path, kernel):
    """{counter: mean} for the kernel whose name contains `kernel`: summaries list a kernel name on a line of its own, followed by
    "COUNTER n=<dispatches> mean[= ]<value>" lines (tools/prof_summary.py and tools/uat_pmc.sh print that shape); a counter seen in
    several passes keeps its first value."""
    out, inside = {}, False
    for line in open(path):
        m = re.match(r"^\s+(\w+)\s+n=\d+\s+mean[= ]\s*([0-9.e+]+)", line)
        if m:
            if inside:
                out.setdefault(m.group(1), float(m.group(2)))
        elif line.strip() and not line.startswith("=="):
            inside = kernel in line
    return out


def first_with_counters(kernel, *names):
    """the first of the summaries that exists and holds FETCH_SIZE and WRITE_SIZE for the kernel (a later round may have profiled a
    workload for time only)"""
    for n in names:
        p = os.path.join(PROF, n)
        if os.path.exists(p):
            m = means(p, kernel)
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                return p
    return None


res = {}
def rounds(pattern):
    """the summaries of every round for one workload, newest first"""
    return [pattern % n for n in range(9, 0, -1)]


p = first_with_counters("scan1090", *rounds("r0%d_final_rocprof_summary.txt"))
if p:
    m = means(p, "scan1090")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        res["1073741824"] = {"traffic_bytes": int(round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)), "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
                             "kernel": "scan1090_kernel", "source": os.path.relpath(p, ROOT),
                             "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on the 1 GiB bench workload; FETCH_SIZE doubled (gfx950 counts half of wide coalesced reads)"}
p = first_with_counters("uat_scan_iq", *rounds("r0%d_uat978_rocprof_summary.txt"))
if p:
    m = means(p, "uat_scan_iq")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        res["uat978:1073741824"] = {"traffic_bytes": int(round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)), "FETCH_SIZE_KB": m["FETCH_SIZE"],
                                    "WRITE_SIZE_KB": m["WRITE_SIZE"], "kernel": "uat_scan_iq_kernel", "source": os.path.relpath(p, ROOT),
                                    "note": "as above, tools/uat_pmc.sh"}
p = first_with_counters("scan2400", *rounds("r0%d_mode2400_rocprof_summary.txt"))
if p:
    m = means(p, "scan2400")
    res["mode2400:1073741824"] = {"traffic_bytes": int(round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)), "FETCH_SIZE_KB": m["FETCH_SIZE"],
                                  "WRITE_SIZE_KB": m["WRITE_SIZE"], "kernel": "scan2400_kernel", "source": os.path.relpath(p, ROOT),
                                  "note": "as above, tools/prof.sh <tag> --rate 24 --no-extras"}
json.dump(res, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
