python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or rccl or rehearsal" > gpurun_out/r04_t20.log 2>&1; tail -2 gpurun_out/r04_t20.log
for i in 1 2; do python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2955$i bench.py --gpus 1 --sharded-step-on-one-rank --steps 200 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sharded_over_plain', d['sharded_over_plain'], 'value', d['value'], 'plain', d['independent_shards_value'], 'ranks_seen', d['ranks_seen'], d['record_transports_agree'], d['serial_step_ms'])"; done
