"""K3 A/B: libdransac variants of the five-point kernels (occupancy / root-finder precision knobs); reports time at 32 and
128 pairs x 1024 samples and the solution recall against the oracle.
  build: python scratch/ab_k3.py --build [names...]     run (GPU box): python scratch/ab_k3.py [names...]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = {'r6ieee': ['-DDR_K3_FAST_DIV=0'], 'r6fdiv': [], 'r6base': [], 'r6asm': ['-DDR_K3_VAR_ASM=1'], 'r6sched': ['-DDR_K3_VAR_SCHED=1'], 'r6estrin': ['-DDR_K3_TASK_ESTRIN=1'], 'r6both': ['-DDR_K3_VAR_ASM=1', '-DDR_K3_TASK_ESTRIN=1'], 'wave': [], 'f32hack_w1': [], 'f32hack_w2': [], 'back1': ['-DDR_K3_BACK_WAVES=1'], 'wave_v1': [], 'wave_v2': [], 'wave_v3': [], 'wave_v4': [], 'nosplit': ['-DDR_ROOT_SPLIT=0'], 'wave_prof': ['-DDR_PROFILE_STAGES'], 'bal_only': ['-DDR_K3_WAVE_ROOTS=0'], 'wave_w2': ['-DDR_K3_WAVES=2'],
            'old': ['-DDR_K3_BALANCED=0', '-DDR_K3_WAVE_ROOTS=0'], 'old_w2': ['-DDR_K3_BALANCED=0', '-DDR_K3_WAVES=2'], 'bal': [], 'bal_w2': ['-DDR_K3_WAVES=2'],
            'old_prof': ['-DDR_K3_BALANCED=0', '-DDR_PROFILE_STAGES'], 'bal_prof': ['-DDR_PROFILE_STAGES'],
            'f32low': ['-DDR_ROOT_F32_LOW=1'], 'f32low_w2': ['-DDR_ROOT_F32_LOW=1', '-DDR_K3_WAVES=2'], 'f32low_prof': ['-DDR_ROOT_F32_LOW=1', '-DDR_PROFILE_STAGES']}
names = [a for a in sys.argv[1:] if not a.startswith('--')] or list(variants)
if '--build' in sys.argv:
    for name in names:
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast',
                               *variants[name], '-o', f'{ROOT}/scratch/libk3_{name}.so', f'{ROOT}/differentiable_ransac_amd/csrc/solve_fivepoint.hip',
                               f'{ROOT}/differentiable_ransac_amd/csrc/dr_core.hip'])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
from oracle import cpu_ref as O
dev = 'cuda'; N, B = 2000, 1024
for P in (32, 128):
    d = synth.batch_two_view(P, N)
    m = d['matches'].to(dev)
    r = ops.gumbel_topk(d['logits'].to(dev), B, 5, 1.0, None, seed=1)
    smp = ops.gather(m, r['idx'], r['y_sel']).reshape(-1, 5, 4).contiguous()
    Bt = smp.shape[0]
    sub = smp[:512].cpu().double()
    Eo, oko, _ = O.nister_5pt(sub)
    ref = {}
    for name in names:
        lib = ctypes.CDLL(f'{ROOT}/scratch/libk3_{name}.so')
        for solver in ('nister5', 'stewenius5'):
            if solver == 'nister5_split' and not hasattr(lib, 'dr_solve_nister5_f32_split'): continue
            models = torch.empty(Bt, 10, 9, device=dev); valid = torch.empty(Bt, 10, device=dev, dtype=torch.uint8)
            cp = lambda t: ctypes.c_void_p(t.data_ptr())
            if solver == 'nister5':
                f = lambda: lib.dr_solve_nister5_f32(cp(smp), None, Bt, 5, cp(models), None, cp(valid), 0, 0, None, None, None)
            elif solver == 'nister5_split':
                wsb = torch.empty(Bt * 88, device=dev, dtype=torch.float64)
                f = lambda: lib.dr_solve_nister5_f32_split(cp(smp), None, Bt, cp(models), None, cp(valid), cp(wsb), None)
            else:
                f = lambda: lib.dr_solve_stewenius5_f32(cp(smp), Bt, cp(models), cp(valid), 0, 0, None, None, None)
            assert f() == 0; torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): f()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 20 * 1e3
            if name.endswith('_prof'):
                buf = (ctypes.c_ulonglong * 16)()
                lib.dr_debug_stage_read_fivepoint(buf); f(); torch.cuda.synchronize(); lib.dr_debug_stage_read_fivepoint(buf)
                print('   cycles/wave by stage:', [int(x) // (Bt // 32) for x in buf[:6]], ' root-search cycles per wave by level 1..10:', [int(x) // (Bt // 32) for x in buf[6:16]])
            Eg = models[:512].reshape(512, 10, 3, 3).cpu().double(); vg = valid[:512].cpu().bool()
            found, total, worst = 0, 0, []
            for i in range(512):
                dist = O.match_solution_sets(Eo[i], oko[i], Eg[i], vg[i])
                total += dist.numel(); found += int((dist < 1e-4).sum()); worst.append(dist)
            w = torch.cat(worst)
            rkey = 'nister5' if solver == 'nister5_split' else solver
            if rkey not in ref: ref[rkey] = (models.clone(), valid.clone())
            rm, rv = ref[rkey]
            same = f'vs first: valid mismatches {(rv != valid).sum().item()}, models max|d| {(rm - models).abs().max().item():.2e}'
            print(f'P={P:4d} {name:16s} {solver:11s}: {us:7.1f} us   valid/sample {valid.float().sum().item()/Bt:.3f}   oracle solutions within 1e-4: '
                  f'{found}/{total}   p99 {w.kthvalue(int(0.99*w.numel())).values:.2e}   {same}', flush=True)
