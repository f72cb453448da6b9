"""c4-size sampler (1 pair x 2048 rows x 50 000 points, k = 3): one-pass kernel vs general two-pass kernel (forced via want_noise is not comparable:
it writes the noise) -- so the general kernel is timed from a -DDR_K1_STREAM=0 build."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = {'stream': [], 'general': ['-DDR_K1_STREAM=0']}
if '--build' in sys.argv:
    for name, flags in variants.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast', *flags,
                               f'-I{ROOT}/include', '-o', f'{ROOT}/scratch/libk1s_{name}.so', f'{ROOT}/differentiable_ransac_amd/csrc/gumbel_topk.hip',
                               f'{ROOT}/differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
dev = 'cuda'
for (P, N, B, k) in ((1, 50000, 2048, 3), (32, 4096, 1024, 5), (8, 20000, 1024, 8)):
    lg = torch.randn(P, N, device=dev)
    idx = torch.empty(P, B, k, device=dev, dtype=torch.int32); ys = torch.empty(P, B, k, device=dev); lse = torch.empty(P, B, device=dev)
    out = {}
    for name in variants:
        lib = ctypes.CDLL(f'{ROOT}/scratch/libk1s_{name}.so')
        cp = lambda t: ctypes.c_void_p(t.data_ptr())
        for mode, (a_y, a_l) in (('index sets', (None, None)), ('soft', (cp(ys), cp(lse)))):
            f = lambda: lib.dr_gumbel_topk_fwd_f32(cp(lg), None, ctypes.c_uint64(7), None, ctypes.c_float(1.0), P, B, N, k, cp(idx), a_y, a_l, None, None, None, None)
            assert f() == 0; torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): f()
            b.record(); torch.cuda.synchronize()
            key = (mode,)
            same = '' if key not in out else f'  idx equal to first variant: {bool(torch.equal(out[key], idx))}'
            out.setdefault(key, idx.clone())
            print(f'P={P} N={N} B={B} k={k} {name:8s} {mode:10s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us{same}', flush=True)
