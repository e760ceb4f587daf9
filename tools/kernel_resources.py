"""VGPR / SGPR / LDS / scratch of every kernel in a built library (from the code object's metadata notes; no GPU needed).
    python tools/kernel_resources.py [libadsb_amd/libadsb_amd.so]"""
import re, subprocess, sys, tempfile, os
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "libadsb_amd", "libadsb_amd.so")
with tempfile.TemporaryDirectory() as d:
    import shutil
    copy = os.path.join(d, "lib.so")
    shutil.copy(lib, copy)  # llvm-objdump --offloading extracts the bundles next to its input
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", copy], stdout=subprocess.DEVNULL)
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(d, f)], text=True)
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk).group(1)
            name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
            name = re.sub(r"\(.*", "", name.replace("adsb_amd::(anonymous namespace)::", "").replace("void ", ""))
            print(f"{name:44s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>7s} scratch {g('private_segment_fixed_size'):>4s}")
