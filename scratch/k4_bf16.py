"""Prototype check: MSAC scoring with the linear forms on the bf16 matrix cores (scratch/k4_bf16.hip) against the
production kernel on the benchmark shape -- scores, mask bits, time.
  build (CPU box):  python scratch/k4_bf16.py --build        run (GPU box):  python scratch/k4_bf16.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
VARIANTS = {'base': [], 'notie': ['-DDR_Q_TIE=0'], 't128w3': ['-DDR_Q_THREADS=128', '-DDR_Q_WAVES=3'],
            't128w3notie': ['-DDR_Q_THREADS=128', '-DDR_Q_WAVES=3', '-DDR_Q_TIE=0'], 'slp': [],
            'noepi': ['-DDR_Q_NOEPI=1'], 'pipe': ['-DDR_Q_PIPE=1'], 'pipe128': ['-DDR_Q_PIPE=1', '-DDR_Q_THREADS=128']}   # noepi: timing decomposition only (wrong results)
NOMASK = os.environ.get('K4_NOMASK') == '1'   # pass masks = NULL: no mask stores, no zero rows
if '--build' in sys.argv:
    for name, flags in VARIANTS.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                               '-ffp-contract=fast', *([] if name == 'slp' else ['-fno-slp-vectorize']), *flags, '-Iinclude',
                               '-o', f'scratch/libk4_bf16_{name}.so', 'scratch/k4_bf16.hip',
                               'differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = 32, 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 9).contiguous(); vflat = valid.reshape(P, -1).contiguous().view(torch.uint8)
M = flat.shape[1]
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
ref_s, ref_m = ops.msac_score(mt, flat, thr, True, valid.reshape(P, -1))
v = valid.reshape(P, -1)
ts = []
for rep in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.msac_score(mt, flat, thr, True, v)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 5)
ts.sort()
print(f'production (incl. output allocation): median {ts[5]*1e3:.1f} us')
for name in (sys.argv[1:] or list(VARIANTS)):
  lib = ctypes.CDLL(os.path.abspath(f'scratch/libk4_bf16_{name}.so'))
  print('variant', name)
  scores = torch.full((P, M), -1.0, device=dev); masks = torch.full((P, M, N), 7, device=dev, dtype=torch.uint8)
  def run():
      rc = lib.dr_msac_score_bf16x3_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                                        ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()),
                                        None if NOMASK else ctypes.c_void_p(masks.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
      assert rc == 0, rc
  run(); torch.cuda.synchronize()
  v = valid.reshape(P, -1)
  ds = ((scores - ref_s).abs() / ref_s.abs().clamp(min=1.0))
  print('scores: max rel diff (valid slots)', float(ds[v].max()), ' invalid slots all zero:', bool((scores[~v] == 0).all()))
  diff = masks != ref_m.view(torch.uint8)
  print('mask bytes differing:', int(diff.sum()), 'of', masks.numel(), ' inliers ref', int(ref_m.sum()), ' values other than 0/1:', int((masks > 1).sum()))
  ts = []
  for rep in range(10):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for _ in range(5): run()
      b.record(); torch.cuda.synchronize()
      ts.append(a.elapsed_time(b) / 5)
  ts.sort()
  print(f'prototype: median {ts[5]*1e3:.1f} us  min {ts[0]*1e3:.1f} us')
