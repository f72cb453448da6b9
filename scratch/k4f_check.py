"""K4 filter kernel (csrc/msac_filter.hip, path 2) against the general kernels (path 1): masks must be bit-identical, scores
equal to f32 rounding; then timing at the benchmark shape, A/B of the build knobs.
  build (CPU box):  python scratch/k4f_check.py --build        run (GPU box):  python scratch/k4f_check.py [--quick]"""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
VARIANTS = {'base': [], 'gs0': ['-DDR_KF_GSHIFT=0'],  
            'storeonly': ['-DDR_KF_STOREONLY=1'],
            **{f'skip{k}': [f'-DDR_KF_SKIP={k}'] for k in (1, 2, 6, 32, 96, 128, 224)},
            'prof': ['-DDR_PROFILE_STAGES']}
if '--build' in sys.argv:
    for name, flags in VARIANTS.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                               '-ffp-contract=fast', '-DDR_KF_STANDALONE', *flags, '-Iinclude', '-o', f'scratch/libk4f_{name}.so',
                               'differentiable_ransac_amd/csrc/msac_filter.hip', 'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
torch.manual_seed(0)
bad = 0


def compare(tag, mt, models, thr, valid=None):
    global bad
    s1, k1 = ops.msac_score(mt, models, thr, True, valid, path=1)
    s2, k2 = ops.msac_score(mt, models, thr, True, valid, path=2)
    torch.cuda.synchronize()
    d = k1 != k2
    nd = int(d.sum())
    nan1, nan2 = torch.isnan(s1), torch.isnan(s2)
    nan_same = bool((nan1 == nan2).all())
    rel = ((s1 - s2).abs() / s1.abs().clamp(min=1.0))[~nan1 & ~nan2]
    mr = float(rel.max()) if rel.numel() else 0.0
    ok = nd == 0 and nan_same and mr < 2e-5
    print(f'{"OK  " if ok else "FAIL"} {tag:46s} mask diffs {nd:8d} / inliers {int(k1.sum()):9d}   score max rel {mr:.2e}  nan pattern same {nan_same}')
    if nd:
        idx = d.nonzero()[:12]
        for p, m, n in idx.tolist():
            print(f'      p {p} m {m} (m%16 {m % 16}) n {n} (wave {n // 256} tile {(n % 256) // 16} q {(n % 16) // 4}): general {int(k1[p, m, n])} filter {int(k2[p, m, n])}')
        print('      missing in filter:', int((k1 & ~k2).sum()), ' extra in filter:', int((k2 & ~k1).sum()))
    if not ok:
        bad += 1
        if mr >= 2e-5:
            w = (((s1 - s2).abs() / s1.abs().clamp(min=1.0)) * (~nan1 & ~nan2)).flatten().argmax()
            print('      worst score:', float(s1.flatten()[w]), float(s2.flatten()[w]), 'flat index', int(w))
    # scores-only call must equal the call with masks
    s3, none = ops.msac_score(mt, models, thr, False, valid, path=2)
    same = torch.equal(torch.nan_to_num(s3, nan=-7.0), torch.nan_to_num(s2, nan=-7.0))
    if not same:
        bad += 1
        print('FAIL   scores-only call differs')
    return s2, k2


# ---- 1. benchmark shape with real five-point models ----
P, N, B = 32, 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 3, 3).contiguous()
v = valid.reshape(P, -1).contiguous()
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
sb, kb = compare('bench shape, five-point models, valid flags', mt, flat, thr, v)
compare('bench shape, no valid flags (eye fillers scored)', mt, flat, thr, None)
sb2, kb2 = ops.msac_score(mt, flat, thr, True, v, path=2)
print('reproducible:', torch.equal(sb, sb2) and torch.equal(kb, kb2))

if '--quick' not in sys.argv:
    # ---- 2. shapes of tests/test_gpu_msac.py that the kernel supports, ragged M, per-pair thresholds ----
    for (n, m, pp) in [(2000, 1024, 1), (2000, 37, 2), (2048, 64, 1), (16, 5, 3), (256, 33, 5), (1024, 160, 7), (1968, 250, 3)]:
        b = synth.batch_two_view(pp, max(n, 8), seed0=100 + n)
        gen = torch.Generator().manual_seed(m)
        md = b['gt_E'][:, None] + 0.05 * torch.randn(pp, m, 3, 3, generator=gen)
        md[:, 0] = b['gt_E']
        if m > 4:
            md[0, 3, 1, 1] = float('nan'); md[0, 4] = float('inf'); md[0, 2] = 0.0
        th = (7.5e-4 * (1 + torch.arange(pp))).to(dev)
        vv = (torch.rand(pp, m, generator=gen) > 0.3).to(dev)
        compare(f'N {n} M {m} P {pp} gt + 0.05 noise, nan/inf/zero models', b['matches'][:, :n].contiguous().to(dev), md.to(dev), th, vv)
        compare(f'N {n} M {m} P {pp} same, models x 2^20', b['matches'][:, :n].contiguous().to(dev), (md * 2.0 ** 20).to(dev), th, None)
    # ---- 3. F matrices on pixel coordinates ----
    b = synth.batch_two_view(4, 2000, seed0=7, pixel=True)
    gen = torch.Generator().manual_seed(5)
    F = b['gt_F'][:, None]
    md = torch.cat((F * (1 + 0.001 * torch.randn(4, 100, 3, 3, generator=gen)), F + F.abs() * 0.05 * torch.randn(4, 156, 3, 3, generator=gen)), 1)
    for th in (0.75, 3.0, 1e-3, 50.0):
        compare(f'F matrices, pixel coordinates, thr {th}', b['matches'].to(dev), md.to(dev), th, None)
    # ---- 4. degenerate inputs ----
    b = synth.batch_two_view(2, 512, seed0=11)
    m2 = b['matches'].clone()
    m2[:, :50, 0] = 0; m2[:, 50:100, 1] = 0; m2[:, 100:150, 2:] = 0; m2[:, 150:160] = 0; m2[:, 160:170] = 1e-30
    E = b['gt_E']
    gen = torch.Generator().manual_seed(11)
    md = torch.cat((E[:, None], E[:, None] * 1e-20, E[:, None] * 1e20, torch.zeros(2, 1, 3, 3), torch.eye(3).expand(2, 1, 3, 3),
                    E[:, None] + 0.01 * torch.randn(2, 50, 3, 3, generator=gen),
                    torch.tensor([1e-8, 0, 0, 0, 1e-8, 0, 0, 0, 1.0]).view(1, 1, 3, 3).expand(2, 1, 3, 3),
                    torch.tensor([0, 0, 1, 0, 0, 0, 0, 0, 0.]).view(1, 1, 3, 3).expand(2, 1, 3, 3),
                    torch.tensor([0, 0, 0, 0, 0, 1, 0, -1, 0.]).view(1, 1, 3, 3).expand(2, 1, 3, 3)), 1).contiguous()
    for th in (7.5e-4, 1e-6, 1e-9, 0.5, 100.0, 0.0):
        compare(f'degenerate points / models, thr {th:g}', m2.to(dev), md.to(dev), th, None)
print('FAILED CHECKS:', bad)

# ---- 5. timing at the benchmark shape ----
def timeit(fn, reps=10, inner=5):
    ts = []
    for _ in range(reps):
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b_.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b_) / inner)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3
M = flat.shape[1]
scores = torch.empty((P, M), device=dev); masks = torch.empty((P, M, N), device=dev, dtype=torch.uint8)
vflat = v.view(torch.uint8)
fm = flat.reshape(P, M, 9)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
from differentiable_ransac_amd import _lib as L
lib = L.lib()
def run_path(path, with_masks=True):
    rc = lib.dr_msac_score_path_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(fm.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                                    ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()),
                                    ctypes.c_void_p(masks.data_ptr()) if with_masks else None, path, st)
    assert rc == 0, rc
for path in (1, 2):
    med, mn = timeit(lambda: run_path(path))
    print(f'path {path}: median {med:.1f} us  min {mn:.1f} us   ({669.49e6 / med / 1e6:.2f} TB/s algorithmic)')
med, mn = timeit(lambda: run_path(2, False))
print(f'path 2 without masks: median {med:.1f} us  min {mn:.1f} us')
med, mn = timeit(lambda: masks.zero_())
print(f'memset of the mask tensor (655 MB): median {med:.1f} us  min {mn:.1f} us')
for name in VARIANTS:
    pth = os.path.abspath(f'scratch/libk4f_{name}.so')
    if not os.path.exists(pth): continue
    vl = ctypes.CDLL(pth)
    def run_v():
        rc = vl.dr_kf_run(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(fm.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                          ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(masks.data_ptr()), st)
        assert rc == 0, rc
    run_v(); torch.cuda.synchronize()
    if name == 'prof':
        buf = (ctypes.c_ulonglong * 16)()
        vl.dr_kf_stage_cycles(buf)          # discard the warm-up call's counts
        run_v(); torch.cuda.synchronize()
        vl.dr_kf_stage_cycles(buf)
        names = ['top (operand reads, model loads issued)', 'filter', 'consume', 'prep_compute', 'barrier wait', 'flush + scores']
        tot = sum(buf[i] for i in range(6))
        nw = 256 * 8
        for i in range(6):
            print(f'   stage {names[i]:42s} {buf[i] / nw:10.0f} cycles per wave  {100.0 * buf[i] / tot:5.1f} %')
        print('   busy cycles (all but the barrier wait) by wave id:', [int(buf[8 + i] / 256) for i in range(8)])
        tr = (ctypes.c_ulonglong * 128)()
        vl.dr_kf_trace(tr)
        t0 = min(tr[16 * wv] for wv in range(8))
        lab = ['start', 'loads issued', 'flush done (A)', 'filter done', 'consume done', 'before stage B', 'stage B done', 'stage A done', 'after barrier', 'filter: matrix+compare done']
        print('   timeline of interval 40, block (0,0), cycles since the first wave entered it; group A = waves 0-3: flush, filter, consume; B = waves 4-7: consume, filter')
        for k in (0, 1, 2, 9, 3, 4, 5, 6, 7, 8):
            print(f'     {lab[k]:32s}', ' '.join(f'{int(tr[16 * wv + k]) - int(t0):7d}' for wv in range(8)))
        print(f'   entries {buf[6]}  batches {buf[7]}  entries per batch {buf[6] / max(1, buf[7]):.1f}  per wave-chunk {buf[6] / (nw * 80):.1f}')
    okv = torch.equal(masks.view(torch.bool), kb)
    med, mn = timeit(run_v)
    def run_v0():
        rc = vl.dr_kf_run(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(fm.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                          ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()), None, st)
        assert rc == 0, rc
    med0, mn0 = timeit(run_v0)
    print(f'variant {name}: median {med:.1f} us  min {mn:.1f} us  masks equal {okv}     without masks: median {med0:.1f} us')
sys.exit(1 if bad else 0)
