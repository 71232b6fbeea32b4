cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench1.json 2> gpurun_out/bench1.err
cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
find $GRAFT_REPO_ROOT/gpurun_out/prof1 -name "*stats*" | head
