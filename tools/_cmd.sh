set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 3 --rehearse-on-one-gpu > gpurun_out/r04_rehearse_n$n.json 2> gpurun_out/r04_rehearse_n$n.err; echo "n=$n rc=$?"; tail -1 gpurun_out/r04_rehearse_n$n.json | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['n_gpus'],d['value'],d['ms_per_step'],d.get('ranks_seen'),d.get('scaling'))"
done
