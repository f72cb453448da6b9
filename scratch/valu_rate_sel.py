import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvalu_sel.so'))
out = torch.empty(1 << 20, device='cuda', dtype=torch.int32); inp = torch.arange(16, device='cuda', dtype=torch.int32)
cyc = torch.zeros(1, device='cuda', dtype=torch.int64)
names = ['cndmask vcc', 'cndmask sgpr pair', 'bfi_b32', 'mov_b32', 'cmp_f64 + cndmask (vcc)', 'cmp_f64 + 4 cndmask (sgpr)', 'and_b32', 'max_f64 (1 chain)']
per = [64, 64, 64, 64, 64, 80, 64, 64]
iters = 500
for mode in range(8):
    f = lambda: lib.run(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(inp.data_ptr()), iters, ctypes.c_void_p(cyc.data_ptr()), 1024, None)
    assert f() == 0; torch.cuda.synchronize()
    f(); torch.cuda.synchronize()
    print(f'{names[mode]:28s}: {int(cyc.item())/(iters*per[mode]):8.2f} clk/inst')
