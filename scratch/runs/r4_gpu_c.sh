#!/bin/bash
# round 4, pass C: new tests; c4 with the screened sampler (and the 4-waves K4r build); the default bench line with the new sub-records
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_c.log; : > $L
echo "== new tests" >> $L
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -8 >> $L
echo "== suite" >> $L
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $L
echo "== K1 at c4, screen off / on" >> $L
timeout 120 python - >> $L 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
it = synth.rigid_pair(0, 50000)
lg = it['logits'][None].to(dev)
for scr in (False, True, False, True):
    for _ in range(5): ops.gumbel_topk(lg, 2048, 3, 1.0, None, 1, soft=False, screen=scr)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(50): ops.gumbel_topk(lg, 2048, 3, 1.0, None, i, soft=False, screen=scr)
    b.record(); torch.cuda.synchronize()
    print('screen', scr, round(a.elapsed_time(b) / 50 * 1e3, 2), 'us per call (incl. the screening launch)')
PY
echo "== c4 step: cur / k4r4" >> $L
for r in 1 2; do for n in cur k4r4; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  DRANSAC_LIB=$lib timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],4), 'ms  residual launch', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3))" >> $L
done; done
echo "== default bench line" >> $L
( time timeout 600 python bench.py > $O/r4_c_bench.json 2> $O/r4_c_bench.err ) 2>> $L
python - >> $L <<'PY'
import json
d = json.load(open('gpurun_out/r4_c_bench.json'))
print('value', round(d['value']/1e6, 2), 'M  ms', round(d['ms_per_step'], 4), 'K4', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 4))
c = d['configs']
for k in sorted(c): print(k, round(c[k]['ms_per_step'], 4), 'ms', round(c[k]['hypotheses_per_s']/1e6, 2), 'M', 'eager', c[k].get('eager_ms_per_step'), 'graph', c[k].get('graph_replay_ms_per_step'))
print('c5 launches', c['c5_train_p32']['launch_ms'])
print('fused', json.dumps(d['fused_driver'])[:900])
print('all_valid', json.dumps(d['k4_all_valid'])[:900])
PY
