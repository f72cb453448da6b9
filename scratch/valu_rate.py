import ctypes, os, torch
lib = ctypes.CDLL(os.path.abspath('scratch/libvalu.so'))
out = torch.empty(1 << 22, device='cuda'); inp = torch.rand(16, device='cuda') + 0.5
cyc = torch.zeros(1, device='cuda', dtype=torch.int64)
names = ['pk_fma 3xVGPR(acc)', 'pk_fma SGPR src', 'fma scalar', 'pk_fma op_sel bcast', 'pk_mul', 'rcp', '1 rcp + 7 pk_fma', '1 fma + 7 pk_fma', 'min_i32', 'pk_fma 3 distinct+dst']
iters = 2000
for threads in (64, 256, 1024):   # waves per SIMD = threads/256 (1 block/CU: 256 blocks)
    for mode in range(10):
        for blocks in (256,):
            f = lambda: lib.run(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(inp.data_ptr()), iters, ctypes.c_void_p(cyc.data_ptr()), blocks, threads, None)
            assert f() == 0; torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            n_inst = iters * 64
            print(f'threads/block={threads:4d} {names[mode]:24s}: {int(cyc.item())/n_inst:6.2f} clk64/inst (wave 0)   wall {ms*1e3:8.1f} us')
