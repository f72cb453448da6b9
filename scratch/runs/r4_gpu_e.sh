#!/bin/bash
# round 4, pass E: f64 backward tests; kernel stats of the long-row sampler with and without the screen
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_e.log; : > $L; R=$PWD
echo "== round-4 tests" >> $L
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -15 >> $L
cat > /tmp/k1_run.py <<'PY'
import torch, sys, os
sys.path.insert(0, os.environ['R'])
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
it = synth.rigid_pair(0, 50000)
lg = it['logits'][None].to(dev)
scr = os.environ['SCR'] == '1'
for i in range(40): ops.gumbel_topk(lg, 2048, 3, 1.0, None, i, soft=False, screen=scr)
torch.cuda.synchronize()
PY
cd /tmp; export TMPDIR=/tmp
for scr in 0 1; do
  R=$R SCR=$scr timeout 200 rocprofv3 --kernel-trace --stats -d $O/r4e_k1_$scr -o k1 -- python /tmp/k1_run.py > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find $O/r4e_k1_$scr -name "*results.db" | head -1) $O/r4e_k1_screen$scr.md "gumbel_topk(50000 x 2048, k = 3, index sets only), screen=$scr, 40 calls" last 30 >> $L 2>&1
  cat $O/r4e_k1_screen$scr.md >> $L
  rm -rf $O/r4e_k1_$scr
done
