import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
variants = {'full': [], 'gram_only': ['-DDR_REFIT_STOP=1'], 'gram_jacobi': ['-DDR_REFIT_STOP=2']}
if '--build' in sys.argv:
    for name, flags in variants.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast', *flags,
                               '-o', f'scratch/librefit_{name}.so', 'differentiable_ransac_amd/csrc/refit.hip', 'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import synth
dev = 'cuda'; P, N = 32, 2000
m = synth.batch_two_view(P, N)['matches'].to(dev).contiguous()
models = torch.empty(P, 10, 9, device=dev); valid = torch.empty(P, 10, device=dev, dtype=torch.uint8)
for name in variants:
    lib = ctypes.CDLL(os.path.abspath(f'scratch/librefit_{name}.so'))
    f = lambda: lib.dr_refit_essential_f32(ctypes.c_void_p(m.data_ptr()), None, P, N, ctypes.c_void_p(models.data_ptr()), ctypes.c_void_p(valid.data_ptr()), None)
    assert f() == 0; torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    print(name, '%.1f us' % (a.elapsed_time(b) / 20 * 1e3))
