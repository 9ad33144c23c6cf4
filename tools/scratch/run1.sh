set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_sources.py tests/test_gpu_parity.py tests/test_gpu_physics.py -x -q 2>&1 | tail -5
python3 tools/scratch/gen_cost.py 5e7 2>&1 | grep -v amdgpu.ids
python3 bench.py --skip-cpu 2>&1 | grep -v amdgpu.ids | cut -c1-900
python3 tools/bench_configs.py C1 C3 C4 C5 2>&1 | grep -v amdgpu.ids
