#!/bin/bash
# round 5: small grids isolate the roots with the wave's idle lanes (multi-point splits) -- tests + the one-pair solver alone + one-pair step
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_solvers.py tests/test_gpu_roots.py tests/test_gpu_round5.py -q --timeout 300 -x 2>&1 | tail -4
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
pair = synth.two_view_pair(3, 2000)
def samples(n):
    r = ops.gumbel_topk(pair['logits'][None].cuda(), n, 5, 1.0, None, 17)
    return ops.gather(pair['matches'][None].cuda(), r['idx'])[0].contiguous()
def t(fn, reps=30, rounds=7):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for _ in range(rounds):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(out)[len(out) // 2]
for n in (1024, 2048, 4096, 8192, 16384, 32768):
    s = samples(n)
    print('samples', n, 'nister', round(t(lambda: ops.solve_nister5(s, path=1)), 1), 'us   stewenius', round(t(lambda: ops.solve_stewenius5(s, path=1)), 1), 'us')
PY
for g in on off; do python bench.py --pairs 1 --graph $g --steps 600 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one pair, graph $g:', round(d['ms_per_step'],4), 'ms')"; done
python scratch/dropin_loop.py 2>&1 | grep "ms per pair"
