cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --one-device --dist-backend gloo --exchange torch --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r04_bench_2rank_one_device.err | tail -1 > gpurun_out/r04_bench_2rank_one_device.json
python3 -c "
import json;b=json.loads(open('gpurun_out/r04_bench_2rank_one_device.json').read().strip().split('\n')[-1]);print(b['value'],b['ms_per_step'],b['launches_per_iteration'],b['collectives_per_iteration'],b['final_rmse_px'],b.get('transport'))"
