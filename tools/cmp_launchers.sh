#!/bin/bash
# bench.py's headline under the three launchers the driver may use (standalone, torch.distributed.run + RCCL, + gloo), twice,
# on one box: ms per step, kernel ms, first / last settle launch.  stderr of every run is kept in gpurun_out/cmp_launchers.err.
# Usage: bash tools/cmp_launchers.sh
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
ERR=gpurun_out/cmp_launchers.err
: > "$ERR"
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['settle_kernel_ms_first_last'])" || { echo "$1: no JSON line, see $ERR"; tail -5 "$ERR"; }; }
for i in 1 2; do
python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-configs 2>>"$ERR" | one standalone
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-configs 2>>"$ERR" | one torchrun_nccl
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-configs --backend gloo 2>>"$ERR" | one torchrun_gloo
done
