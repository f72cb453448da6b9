#!/bin/bash
mkdir -p gpurun_out/r3k
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3k/pytest.log)
tail -4 gpurun_out/r3k/pytest.log
timeout 200 python bench.py --mode train --steps 200 --warmup 5 --segments 3 > gpurun_out/r3k/train.json 2> gpurun_out/r3k/train.err
python -c "
import json; r=json.load(open('gpurun_out/r3k/train.json')); print('train graph', r['value'], r['ms_per_step'], r['segments']['ms_per_step'])"
timeout 200 python bench.py --mode train --graph off --steps 200 --warmup 5 --segments 3 > gpurun_out/r3k/train_eager.json 2> gpurun_out/r3k/train_eager.err
python -c "
import json; r=json.load(open('gpurun_out/r3k/train_eager.json')); print('train eager', r['value'], r['ms_per_step'])"
python - <<'PY'
import torch, time
from differentiable_ransac_amd import ops, synth
dev='cuda'
P,N,M=32,2000,1024
d=synth.batch_two_view(P,N)
mt=d['matches'].to(dev); mask=d['inliers'].to(dev)
models=(d['gt_E'][:,None]+0.05*torch.randn(P,M,3,3)).to(dev).requires_grad_(True)
def t(fn,reps=50):
    fn(); torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps*1e3
from differentiable_ransac_amd import _lib as L
from differentiable_ransac_amd._lib import ptr, stream, c_int
sums=torch.empty(P,M,device=dev); gm=torch.empty(P,M,9,device=dev); gs=torch.ones(P,M,device=dev)
mk=mask.view(torch.uint8); md=models.detach().contiguous()
print('episym fwd us', t(lambda: L.call('dr_episym_fwd_f32', ptr(mt), ptr(mk), ptr(md), ptr(None), c_int(P), c_int(M), c_int(N), ptr(sums), stream())))
print('episym bwd us', t(lambda: L.call('dr_episym_bwd_f32', ptr(mt), ptr(mk), ptr(md), ptr(None), ptr(gs), c_int(P), c_int(M), c_int(N), ptr(gm), stream())))
PY
