"""Cycles per pass (16 tiles) of the K4 filter's inner loop, in isolation.  build: --build; run on the GPU box."""
import ctypes, os, subprocess, sys
V = {f'm{m}g{g}': [f'-DMODE={m}', f'-DG={g}'] for m in (0, 1, 2) for g in (1, 2, 4)}
if '--build' in sys.argv:
    for k, f in V.items():
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', *f, '-o', f'scratch/libprobe_{k}.so', 'scratch/mfma_probe.hip'])
    sys.exit(0)
import torch
inp = torch.rand(2048, device='cuda') - 0.5
out = torch.empty(256 * 512, device='cuda', dtype=torch.int32)
cyc = torch.zeros(1, device='cuda', dtype=torch.int64)
reps = 2000
for k in V:
    lib = ctypes.CDLL(os.path.abspath(f'scratch/libprobe_{k}.so'))
    for threads in (256, 512):
        lib.run_probe(ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()), reps, ctypes.c_void_p(cyc.data_ptr()), threads, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.run_probe(ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()), reps, ctypes.c_void_p(cyc.data_ptr()), threads, None)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        print(f'{k} (mode 0 both / 1 matrix / 2 vector; G tiles per stage) waves/SIMD {threads // 256}: {int(cyc[0]) / reps:8.1f} cycles per pass of 16 tiles (wave 0), kernel {ms * 1e3:.0f} us -> {ms * 1e-3 / reps * 1e9:.0f} ns per pass')
