"""Builds a profiling variant of the library (-DDR_PROFILE_STAGES) and prints per-stage cycle totals of the Nister kernel."""
import ctypes, os, subprocess, sys, glob
sys.path.insert(0, '.')
import torch
src = sorted(glob.glob('differentiable_ransac_amd/csrc/*.hip'))
out = 'gpurun_out/libdransac_prof.so'
os.makedirs('gpurun_out', exist_ok=True)
extra = [a for a in sys.argv[1:] if a.startswith('-D')]
tag = ''.join(a.replace('-D', '_').replace('=', '') for a in extra)
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast',
                           '-DDR_PROFILE_STAGES', *extra, '-o', f'scratch/libdransac_prof{tag}.so', *src])
    sys.exit(0)
import differentiable_ransac_amd._lib as L
L.LIB_PATH = os.path.abspath(f'scratch/libdransac_prof{tag}.so')
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = int(os.environ.get('K3_PAIRS', 32)), 2000, 1024
PATH = int(os.environ.get('K3_PATH', 1))
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'], r['y_sel'])
lib = L.lib()
buf = (ctypes.c_ulonglong * 32)()
for name, fn in (('nister', lambda x: ops.solve_nister5(x, path=PATH)),):
    fn(smp); torch.cuda.synchronize()
    lib.dr_debug_stage_read_fivepoint(buf)
    fn(smp); torch.cuda.synchronize()
    lib.dr_debug_stage_read_fivepoint(buf)
    tot = sum(buf)
    print(name, 'path', PATH, 'waves (32-sample units)', P * B // 32, 'cycles/wave by stage:', [int(b) // (P * B // 32) for b in buf[:10]] + ['final', int(buf[11]) // (P * B // 32)] + [round(int(buf[i]) / (P * B // 32), 2) for i in (13, 14, 15)], 'max isolation iterations of a wave', int(buf[10]), '| final stage: start', int(buf[16]) // (P * B // 32), 'precheck', int(buf[17]) // (P * B // 32), 'steps', int(buf[18]) // (P * B // 32), 'verify+store', int(buf[19]) // (P * B // 32), 'identity', int(buf[20]) // (P * B // 32), '| degenerate lanes', int(buf[21]), 'waves with one', int(buf[22]), 'by leading coefficient (lane x member)', int(buf[23]), 'by remainder', int(buf[24]))
