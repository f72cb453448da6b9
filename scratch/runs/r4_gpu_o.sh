#!/bin/bash
# round 4, pass O: five-point kernel with fewer samples per block on small grids: suite, one-pair call, headline unchanged
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_o.log; : > $L
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 >> $L
for g in off on; do timeout 120 python bench.py --pairs 1 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras --graph $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=1 graph=$g', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s  K4', round(d['roofline']['avg_launch_ms'],4))" >> $L; done
for p in 2 4 8 16; do timeout 120 python bench.py --pairs $p --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$p', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s')" >> $L; done
timeout 100 python - >> $L 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
dev='cuda'
d=synth.batch_two_view(1,2000)
for nsmp in (1024, 2048, 4096, 8192, 16384):
    r=ops.gumbel_topk(d['logits'].to(dev),nsmp,5,1.0,None,seed=1,soft=False)
    smp=ops.gather(d['matches'].to(dev), r['idx'])
    for _ in range(10): ops.solve_nister5(smp)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): ops.solve_nister5(smp)
    b.record(); torch.cuda.synchronize()
    print('nister', nsmp, 'samples', round(a.elapsed_time(b)/50*1e3,1), 'us per call')
PY
bash scratch/ab_step.sh cur >> $L 2>&1
