import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvalu_f64.so'))
out = torch.empty(1 << 20, device='cuda', dtype=torch.float64); inp = torch.rand(16, device='cuda', dtype=torch.float64) * 0.5 + 0.25
cyc = torch.zeros(1, device='cuda', dtype=torch.int64)
names = ['fma_f64 8 chains', 'fma_f64 1 chain', 'fma_f64 2 chains', 'fma_f64 4 chains', 'add_f64', 'mul_f64', 'rcp_f64', 'cndmask_b32', 'cmp_lt_f64', 'fma_f32 1 chain', 'fma_f32 8 chains', 'bisect step d=10 (x8)']
iters = 500
for blocks in (1024, 2048):    # one / two waves per SIMD
    for mode in range(12):
        f = lambda: lib.run(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(inp.data_ptr()), iters, ctypes.c_void_p(cyc.data_ptr()), blocks, None)
        assert f() == 0; torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        n = iters * (8 if mode == 11 else 64)
        print(f'blocks={blocks:5d} {names[mode]:24s}: {int(cyc.item())/n:8.2f} clk/inst (wave 0)   wall {a.elapsed_time(b)*1e3:8.1f} us  -> {a.elapsed_time(b)*1e-3/n*1e9:6.2f} ns/inst')
