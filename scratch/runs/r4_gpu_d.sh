#!/bin/bash
# round 4, pass D: why the screened sampler does not pay (kernel stats), the anomalous gather_bwd figure of the train split
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_d.log; : > $L; R=$PWD
cat > /tmp/k1_run.py <<'PY'
import torch, sys, os
sys.path.insert(0, os.environ['R'])
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
it = synth.rigid_pair(0, 50000)
lg = it['logits'][None].to(dev)
print('logits mean/std/max', float(lg.mean()), float(lg.std()), float(lg.max()))
for scr in (False, True):
    for i in range(30): ops.gumbel_topk(lg, 2048, 3, 1.0, None, i, soft=False, screen=scr)
torch.cuda.synchronize()
PY
cd /tmp; export TMPDIR=/tmp
R=$R timeout 200 rocprofv3 --kernel-trace --stats -d $O/r4d_k1 -o k1 -- python /tmp/k1_run.py >> $L 2>&1
python - >> $L <<PY
import glob, csv
for f in glob.glob("$O/r4d_k1/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print(row['Name'][:90], row['Calls'], row['AverageNs'], row['MinNs'])
PY
rm -rf $O/r4d_k1
cd $R
echo "== train split per rep" >> $L
timeout 200 python - >> $L 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
import bench
from differentiable_ransac_amd import _lib as L
dev = torch.device('cuda', 0)
w = dict(bench.WORKLOADS["c2"], pairs=32)
step, _ = bench.make_step(w, dev, mode="train")
for _ in range(20): step()
torch.cuda.synchronize()
orig = L.call; rec = []
def call(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record(); rec.append((name, e0, e1))
L.call = call
for _ in range(6): step()
torch.cuda.synchronize()
L.call = orig
per = {}
for n, a, b in rec: per.setdefault(n, []).append(round(a.elapsed_time(b) * 1e3, 1))
for n, v in per.items(): print(n, v)
# the same with the device kept busy: 4 steps queued behind a long kernel
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
rec.clear(); L.call = call
for _ in range(20): x.zero_()
for _ in range(6): step()
torch.cuda.synchronize(); L.call = orig
per = {}
for n, a, b in rec: per.setdefault(n, []).append(round(a.elapsed_time(b) * 1e3, 1))
print('-- device kept busy ahead of the host')
for n, v in per.items(): print(n, v)
PY
