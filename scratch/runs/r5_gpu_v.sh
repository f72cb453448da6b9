#!/bin/bash
# round 5: K4 on tiny grids -- 16- / 8- / 4-slot halves (DR_K4_SMALL_TILE) at 1, 2, 3 pairs: replayed step and the scoring launch
mkdir -p gpurun_out/r5v
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5v
cd $R
timeout 300 python -m pytest tests/test_gpu_msac.py -q --timeout 300 -x 2>&1 | tail -2
for P in 1 2 3; do
  for ts in 16 8 4; do
    DR_K4_SMALL_TILE=$ts python bench.py --pairs $P --graph on --steps 600 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs $P slots $ts: replayed step', round(d['ms_per_step'],4), 'ms   scoring launch (eager, HIP events)', round(d['roofline']['avg_launch_ms']*1e3,2), 'us')"
  done
done
for ts in 16 8 4; do
DR_K4_SMALL_TILE=$ts python - <<PY
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
d = synth.batch_two_view(1, 2000)
m, lg = d['matches'].cuda(), d['logits'].cuda()
idx, smp = ops.gumbel_topk_gather(m, lg, 1024, 5, 1.0, 3)
E, v = ops.solve_nister5(smp)
thr = 7.5e-4
for _ in range(20): ops.msac_score(m, E.reshape(1, -1, 3, 3), thr, valid=v.reshape(1, -1))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(7):
    e0.record()
    for _ in range(50): ops.msac_score(m, E.reshape(1, -1, 3, 3), thr, valid=v.reshape(1, -1))
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
print('isolated back-to-back, one pair, slots $ts:', round(sorted(ts)[3], 2), 'us')
PY
done
