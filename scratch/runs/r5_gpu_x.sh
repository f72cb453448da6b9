#!/bin/bash
# round 5: essential-refit kernel with the two-phase Jacobi rounds -- tests, the kernel alone (1 / 128 pairs), the step with the refit
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_round3.py tests/test_gpu_drivers.py tests/test_gpu_round5.py -q --timeout 300 -x 2>&1 | tail -3
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
for P in (1, 128):
    d = synth.batch_two_view(P, 2000)
    m = d['matches'].cuda()
    for _ in range(10): ops.refit_essential(m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        e0.record()
        for _ in range(30): ops.refit_essential(m)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    print('refit_essential alone, pairs', P, round(sorted(ts)[3], 1), 'us')
PY
python scratch/refit_step.py 2>&1 | grep "refit="
python scratch/dropin_loop.py 2>&1 | grep -v amdgpu.ids | tail -4
