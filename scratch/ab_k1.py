"""K1 timing experiments (wrong noise in all but `base`): how much of the sampler is Philox, how much the two logarithms?
  build: python scratch/ab_k1.py --build      run (GPU box): python scratch/ab_k1.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
variants = {'base': [], 'general': ['-DDR_K1_FAST=0'], 'nolog': ['-DDR_K1_NOISE_EXPERIMENT=1'], 'onelog': ['-DDR_K1_NOISE_EXPERIMENT=2'],
            'philox7': ['-DDR_PHILOX_ROUNDS=7'], 'philox0': ['-DDR_PHILOX_ROUNDS=0'],
            'philox0_nolog': ['-DDR_PHILOX_ROUNDS=0', '-DDR_K1_NOISE_EXPERIMENT=1']}
if '--build' in sys.argv:
    for name, flags in variants.items():
        if [a for a in sys.argv[1:] if not a.startswith('--')] and name not in sys.argv: continue
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast', *flags,
                               '-Iinclude', '-o', f'scratch/libk1_{name}.so', 'differentiable_ransac_amd/csrc/gumbel_topk.hip', 'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import synth
dev = 'cuda'; P, N, B, k = int(os.environ.get('AB_K1_PAIRS', 32)), 2000, 1024, 5
lg = synth.batch_two_view(P, N)['logits'].to(dev)
idx = torch.empty(P, B, k, device=dev, dtype=torch.int32); ys = torch.empty(P, B, k, device=dev); lse = torch.empty(P, B, device=dev)
names = [a for a in sys.argv[1:] if not a.startswith('--')] or list(variants)
for name in names:
    lib = ctypes.CDLL(os.path.abspath(f'scratch/libk1_{name}.so'))
    cp = lambda t: ctypes.c_void_p(t.data_ptr())
    for mode, (a_y, a_l) in (('index sets only (test mode)', (None, None)), ('with soft-max statistics (train mode)', (cp(ys), cp(lse)))):
        f = lambda: lib.dr_gumbel_topk_fwd_f32(cp(lg), None, ctypes.c_uint64(7), None, ctypes.c_float(1.0), P, B, N, k, cp(idx), a_y, a_l, None, None, None, None)
        assert f() == 0; torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        print(f'{name:14s} {mode:40s} {a.elapsed_time(b) / 20 * 1e3:7.1f} us')
