python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench6.json 2> gpurun_out/r04_bench6.err
python3 bench.py --steps 20 --warmup 5 --no-extras --cpu-buffers 0 > gpurun_out/r04_bench6b.json 2>/dev/null
python3 bench.py --steps 200 --warmup 10 --no-extras --cpu-buffers 0 > gpurun_out/r04_bench6c.json 2>/dev/null
python3 bench.py --workload uat978 --steps 20 --warmup 5 > gpurun_out/r04_bench6_uat.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("r04_bench6", "r04_bench6b", "r04_bench6c"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, {k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_first_100"), d.get("mode_2400", {}).get("ms_per_step"), d.get("mode_2400", {}).get("kernel_ms"), d.get("uat978", {}).get("ms_per_step"), d.get("host_resolve_ms_rank0"), d.get("decoded_msgs_per_s"))
d = json.loads(open("gpurun_out/r04_bench6_uat.json").read().strip().splitlines()[-1])
print("uat", d["value"], d["ms_per_step"], d["ms_per_step_serial"], d["roofline"]["kernel_ms"], d["demod_kernel_ms"])
d = json.loads(open("gpurun_out/r04_bench6.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("mode_2400", {}).get("transmitted_frames_recovered")), json.dumps(d.get("end_to_end")), json.dumps(d.get("cpu_baseline")))
PY
