import ctypes, os, sys, torch
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
lib = ctypes.CDLL(os.path.abspath('scratch/libk4mfma.so'))
dev = 'cuda'; P, N, B = 32, 2000, 1024
data = synth.batch_two_view(P, N)
mt = data['matches'].to(dev).contiguous()
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
models, valid = ops.solve_nister5(ops.gather(mt, r['idx'], r['y_sel']))
flat = models.reshape(P, -1, 9).contiguous(); vflat = valid.reshape(P, -1)
M = flat.shape[1]; Npad = (N + 15) // 16 * 16
thr = torch.full((P,), 7.5e-4, device=dev)
phi = torch.empty(P, 24, Npad, device=dev); coef = torch.empty(P, M, 24, device=dev); sc = torch.empty(P, M, device=dev)
cp = lambda t: ctypes.c_void_p(t.data_ptr())
def prep():
    assert lib.run_features(cp(mt), P, N, Npad, cp(phi), None) == 0
    assert lib.run_coeffs(cp(flat), P, M, cp(coef), None) == 0
def score(mode=0):
    assert lib.run_score(cp(phi), cp(coef), cp(thr), P, M, Npad, cp(sc), None, mode) == 0
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
prep(); score()
ref, _ = ops.msac_score(mt, flat, thr, want_masks=False)            # all slots evaluated
err = (sc - ref).abs() / ref.abs().clamp(min=1.0)
print('score rel err (vs VALU kernel): max %.3e  p99 %.3e  median %.3e' % (err.max(), err.flatten().kthvalue(int(0.99 * err.numel())).values, err.median()))
v = vflat
print('  on valid slots only: max %.3e' % err[v].max())
print('prep %.1f us   score_mfma (all %d slots) %.1f us   VALU kernel all slots, no masks %.1f us, valid slots only %.1f us' % (
    t(prep), M, t(score), t(lambda: ops.msac_score(mt, flat, thr, want_masks=False)), t(lambda: ops.msac_score(mt, flat, thr, want_masks=False, valid=vflat))))

print('mfma only %.1f us   epilogue only %.1f us' % (t(lambda: score(1)), t(lambda: score(2))))
