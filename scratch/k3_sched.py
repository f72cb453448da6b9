"""A/B of the root-finder schedule: builds libdransac variants with -DDR_ROOT_* and reports K3 time + solution recall vs the oracle."""
import ctypes, os, subprocess, sys, glob
sys.path.insert(0, '.')
variants = {'base': [], 'low43': ['-DDR_ROOT_BIS_LOW=4', '-DDR_ROOT_NEWT_LOW=3'], 'low43_last75': ['-DDR_ROOT_BIS_LOW=4', '-DDR_ROOT_NEWT_LOW=3', '-DDR_ROOT_BIS_LAST=7', '-DDR_ROOT_NEWT_LAST=5'],
            'low32_last64': ['-DDR_ROOT_BIS_LOW=3', '-DDR_ROOT_NEWT_LOW=2', '-DDR_ROOT_BIS_LAST=6', '-DDR_ROOT_NEWT_LAST=4']}
if '--build' in sys.argv:
    for name, flags in variants.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast', *flags,
                               '-o', f'scratch/libk3_{name}.so', 'differentiable_ransac_amd/csrc/solve_fivepoint.hip', 'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
from oracle import cpu_ref as O
dev = 'cuda'; P, N, B = 32, 2000, 1024
d = synth.batch_two_view(P, N)
m = d['matches'].to(dev)
r = ops.gumbel_topk(d['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(m, r['idx'], r['y_sel']).reshape(-1, 5, 4).contiguous()
Bt = smp.shape[0]
# oracle on a subset
sub = smp[:512].cpu().double()
Eo, oko, _ = O.nister_5pt(sub)
for name in variants:
    lib = ctypes.CDLL(os.path.abspath(f'scratch/libk3_{name}.so'))
    models = torch.empty(Bt, 10, 9, device=dev); valid = torch.empty(Bt, 10, device=dev, dtype=torch.uint8)
    f = lambda: lib.dr_solve_nister5_f32(ctypes.c_void_p(smp.data_ptr()), None, Bt, 5, ctypes.c_void_p(models.data_ptr()), None, ctypes.c_void_p(valid.data_ptr()), 0, 0, None, None, None)
    assert f() == 0; torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    Eg = models[:512].reshape(512, 10, 3, 3).cpu().double(); vg = valid[:512].cpu().bool()
    found, total, worst = 0, 0, []
    for i in range(512):
        dist = O.match_solution_sets(Eo[i], oko[i], Eg[i], vg[i])
        total += dist.numel(); found += int((dist < 1e-4).sum()); worst.append(dist)
    w = torch.cat(worst)
    print(f'{name:14s}: {us:7.1f} us   valid/sample {valid.float().sum().item()/Bt:.3f}   oracle solutions recovered within 1e-4: {found}/{total}   p99 {w.kthvalue(int(0.99*w.numel())).values:.2e}')
