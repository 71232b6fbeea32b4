set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
lscpu | grep -E "Model name|^CPU\(s\)" 
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=8 2>&1 | tail -30 > gpurun_out/first_parity.log
cat gpurun_out/first_parity.log
python - <<'PY' 2>&1 | tee gpurun_out/first_counts.log
import time, numpy as np, hesaff_amd
from hesaff_amd.synth import band_noise_image
from tests import _oracle
img = band_noise_image(1080, 1920, 1235)
t=time.time(); o=_oracle.OracleRun(_oracle.gray_from_u8(img)); t_or=time.time()-t
c = hesaff_amd.HesaffContext()
c.set_profiling(2)
t=time.time(); (nh, keys), = c.detect_batch([img]); t1=time.time()-t
t=time.time(); (nh, keys), = c.detect_batch([img]); t2=time.time()-t
print("oracle", o.n_hessian, o.n_keys, "%.2fs"%t_or, "gpu", nh, len(keys), "first %.3fs second %.3fs"%(t1,t2))
tm=c.timings()
print({k:getattr(tm,k) for k,_ in tm._fields_})
PY
