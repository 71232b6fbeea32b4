# usage: bash scripts/gpu_timeline.sh <tag> : kernel trace of one overlapped bench run + a text timeline of the last step
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch ${BATCH:-32} --no-cpu-baseline --no-host-path > /dev/null 2>&1)
python3 scripts/timeline.py gpurun_out/tl_$1/p_kernel_trace.csv ${SLICE_MS:-5}
find gpurun_out/tl_$1 -name "*kernel_trace.csv" -delete; find gpurun_out/tl_$1 -name "*.db" -delete
