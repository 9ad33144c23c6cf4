set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/gen_cost2 -- python3 $R/tools/scratch/gen_cost.py 1e7 > $R/gpurun_out/gen_cost2.log 2>&1
