mkdir -p gpurun_out/r2v
export B4D_BENCH_BACKEND=gloo
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2v/bench2.json 2> gpurun_out/r2v/bench2.err; echo "rc $?"
tail -5 gpurun_out/r2v/bench2.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r2v/bench2.json").read().strip().splitlines()[-1])
print(l["n_gpus"], l["value"]); print(json.dumps(l["secondary"], indent=1)[:1500])
PY
