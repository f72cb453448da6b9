import ctypes, os, torch
lib = ctypes.CDLL(os.path.abspath('scratch/libclk.so'))
out = torch.empty(1 << 22, device='cuda'); inp = torch.rand(16, device='cuda') + 0.5
ticks = torch.zeros(1, device='cuda', dtype=torch.int64)
names = ['4 mfma_f32_16x16x4 / iter', '16 v_pk_fma_f32 / iter', 'both (mfma block, then pk block)', '24 v_fma_f32 / iter', '4 x (mfma + 6 v_fma_f32) interleaved']
for blocks in (256, 1024):      # 1 or 4 waves per SIMD
    for mode in range(5):
        iters = 20000
        f = lambda: lib.run(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(inp.data_ptr()), iters, ctypes.c_void_p(ticks.data_ptr()), blocks, None)
        assert f() == 0; torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3
        tk = int(ticks.item())
        print(f'blocks={blocks:5d} {names[mode]:28s}: wall {us:9.1f} us  ticks {tk:10d}  -> {tk/us:7.1f} ticks/us;  per iter {us*1e3/iters:7.2f} ns = {tk/iters:7.1f} ticks (wave 0)')
