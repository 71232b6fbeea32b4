# usage (through gpurun): bash scripts/gpu_variants.sh "<name>:<lib>[:ENV=val,...]" ...   per-kernel serial timings of tuning-build variants
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=""
  if [ "$rest" != "$lib" ]; then envs=$(echo ${rest#*:} | tr ',' ' '); fi
  echo "=== $name ($lib $envs)"
  env $envs bash scripts/gpu_kernels.sh var_$name 32 $GRAFT_REPO_ROOT/hesaff_amd/variants/$lib.so | head -12
done
