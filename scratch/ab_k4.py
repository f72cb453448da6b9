"""A/B harness for K4 variants: compiles msac_score.hip with -DDR_K4_VARIANT=<v> into separate shared objects and times
dr_msac_score_f32 on identical inputs, interleaved.   python scratch/ab_k4.py --build 0 1 2   (CPU box)
                                                        python scratch/ab_k4.py 0 1 2           (GPU box)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
args = [a for a in sys.argv[1:] if a != '--build']
variants = [int(a) for a in args] or [0]
if '--build' in sys.argv:
    for v in variants:
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                               '-ffp-contract=fast', f'-DDR_K4_VARIANT={v % 100}', *(['-DDR_K4_NOREDUCE'] if v // 100 == 7 else []), *(['-DDR_K4_BPERMUTE_SUM'] if v // 100 == 8 else []), *([f'-DDR_K4_TILE16={32 * (v // 1000)}'] if v >= 1000 else []), *(['-fno-slp-vectorize'] if v == 9 else []), '-o', f'scratch/libk4_v{v}.so',
                               'differentiable_ransac_amd/csrc/msac_score.hip', 'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = 32, 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'], r['y_sel'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 9).contiguous(); vflat = valid.reshape(P, -1).contiguous().view(torch.uint8)
M = flat.shape[1]
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
scores = torch.empty(P, M, device=dev); masks = torch.empty(P, M, N, device=dev, dtype=torch.uint8)
libs = {v: ctypes.CDLL(os.path.abspath(f'scratch/libk4_v{v}.so')) for v in variants}
def run(lib, with_masks, with_valid):
    lib.dr_msac_score_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(flat.data_ptr()),
                          ctypes.c_void_p(vflat.data_ptr()) if with_valid else None, ctypes.c_void_p(thr.data_ptr()), P, M, N,
                          ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(masks.data_ptr()) if with_masks else None,
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
ref = None
for mode in ((True, True), (False, True), (True, False)):
    res = {v: [] for v in variants}
    for rep in range(12):
        for v in variants:
            run(libs[v], *mode); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): run(libs[v], *mode)
            b.record(); torch.cuda.synchronize()
            res[v].append(a.elapsed_time(b) / 5)
    for v in variants:
        run(libs[v], *mode); torch.cuda.synchronize()
        chk = (float(scores.nan_to_num().sum()), int(masks.sum()) if mode[0] else -1)
        t = sorted(res[v])
        print(f'masks={mode[0]} valid={mode[1]} variant {v}: median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us  check {chk}')
# padded mask rows (stride 2048): every row store is a whole number of 128-byte lines
mp = torch.empty(P, M, 2048, device=dev, dtype=torch.uint8)
for v in variants:
    lib = libs[v]
    if not hasattr(lib, 'dr_msac_score_strided_f32'): continue
    def runs():
        return lib.dr_msac_score_strided_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                              ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()),
                              ctypes.c_void_p(mp.data_ptr()), 2048, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert runs() == 0; torch.cuda.synchronize()
    ts = []
    for rep in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): runs()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 5)
    ts.sort()
    print(f'variant {v}: stride-2048 masks median {ts[6]*1e3:.1f} us min {ts[0]*1e3:.1f} us check {(float(scores.nan_to_num().sum()), int(mp[..., :N].sum()), int(mp[..., N:].sum()))}')
# store-only floor: all slots invalid -> the kernel only zero-fills the 655 MB mask tensor
vz = torch.zeros_like(vflat)
for v in variants:
    lib = libs[v]
    def run0():
        lib.dr_msac_score_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(vz.data_ptr()),
                              ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()),
                              ctypes.c_void_p(masks.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    run0(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run0()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20
    print(f'variant {v}: zero-fill only {t*1e3:.1f} us = {masks.numel()/t/1e9:.0f} GB/s')
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
masks.zero_(); torch.cuda.synchronize(); a.record()
for _ in range(20): masks.zero_()
b.record(); torch.cuda.synchronize(); t = a.elapsed_time(b) / 20
print(f'torch memset of the mask tensor: {t*1e3:.1f} us = {masks.numel()/t/1e9:.0f} GB/s')
