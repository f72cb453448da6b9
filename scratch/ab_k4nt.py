"""A/B of general-scoring-kernel build variants (tile size, mask-store policy) at the benchmark shape (K4_PAIRS pairs).
  build (CPU box): python scratch/ab_k4nt.py --build      run (GPU box): python scratch/ab_k4nt.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
VARIANTS = {'base': [], 'h1': ['-DDR_K4_HALVES=1']}
if '--build' in sys.argv:
    for name, flags in VARIANTS.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast',
                               *flags, '-Iinclude', '-o', f'scratch/libk4nt_{name}.so', 'differentiable_ransac_amd/csrc/msac_score.hip',
                               'differentiable_ransac_amd/csrc/msac_filter.hip', 'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = int(os.environ.get('K4_PAIRS', '32')), 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 9).contiguous(); vflat = valid.reshape(P, -1).contiguous().view(torch.uint8)
M = flat.shape[1]
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
scores = torch.empty(P, M, device=dev); masks = torch.empty(P, M, N, device=dev, dtype=torch.uint8)
ref = None
libs = {k: ctypes.CDLL(os.path.abspath(f'scratch/libk4nt_{k}.so')) for k in VARIANTS}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(lib):
    rc = lib.dr_msac_score_path_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                                    ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(masks.data_ptr()), 1, st)
    assert rc == 0
res = {k: [] for k in VARIANTS}
for rep in range(15):
    for k, lib in libs.items():
        run(lib); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(lib)
        b.record(); torch.cuda.synchronize()
        res[k].append(a.elapsed_time(b) / 5)
for k, lib in libs.items():
    run(lib); torch.cuda.synchronize()
    chk = (float(scores.nan_to_num().sum()), int(masks.sum()))
    t = sorted(res[k])
    print(f'general kernel, variant {k}: median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us  check {chk}')
