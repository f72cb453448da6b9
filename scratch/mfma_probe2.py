import ctypes, os, subprocess, sys
V = {f'm{m}': [f'-DMODE={m}'] for m in range(5)}
NAMES = ['16x16x32 f16, separate live C', '16x16x32 f16, C = 0', '16x16x32 bf16, C = 0', '32x32x16 f16, C = 0 (16 per pass)', '16x16x16 f16 (K = 16), C = 0']
if '--build' in sys.argv:
    for k, f in V.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', *f, '-o', f'scratch/libprobe2_{k}.so', 'scratch/mfma_probe2.hip'])
    sys.exit(0)
import torch
inp = torch.rand(2048, device='cuda') - 0.5
out = torch.empty(256 * 512, device='cuda', dtype=torch.int32)
cyc = torch.zeros(1, device='cuda', dtype=torch.int64)
reps = 2000
for i, k in enumerate(V):
    lib = ctypes.CDLL(os.path.abspath(f'scratch/libprobe2_{k}.so'))
    for threads in (256, 512):
        lib.run_probe(ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()), reps, ctypes.c_void_p(cyc.data_ptr()), threads, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.run_probe(ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()), reps, ctypes.c_void_p(cyc.data_ptr()), threads, None)
        b.record(); torch.cuda.synchronize()
        ns = a.elapsed_time(b) * 1e6 / reps
        nm = 16 if i == 3 else 32
        print(f'{NAMES[i]:36s} waves/SIMD {threads // 256}: wave 0 {int(cyc[0]) / reps / nm:6.1f} cycles per matrix instruction; kernel {ns:7.1f} ns per pass = {ns / (nm * threads // 256):6.2f} ns per instruction per SIMD')
