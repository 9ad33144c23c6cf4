one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['settle_kernel_ms_first_last'])"; }
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --skip-cpu 2>/dev/null | one standalone
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu 2>/dev/null | one torchrun_nccl
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --backend gloo 2>/dev/null | one torchrun_gloo
done
