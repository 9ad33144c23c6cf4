set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/det_prof -- python3 $R/tools/bench_detector.py > $R/gpurun_out/det_prof.log 2>&1
