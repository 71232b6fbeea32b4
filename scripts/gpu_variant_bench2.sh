# usage (through gpurun): bash scripts/gpu_variant_bench2.sh "<ENV=..> <variant>" ...   -- step time of the bench (B = 128) per variant library and environment
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  set -- $spec
  v=${@: -1}
  envs="${@:1:$#-1}"
  env $envs HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch 128 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s step %.1f img/s %.1f' % ('$spec', d['ms_per_step'], d['images_per_s']))"
done
