import ctypes, os, torch
lib = ctypes.CDLL(os.path.abspath('scratch/libvalu2.so'))
out = torch.empty(1 << 22, device='cuda'); inp = torch.rand(16, device='cuda') + 0.5
names = ['v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32', 'v_mul_u32_u24', 'v_log_f32', 'v_exp_f32', 'v_fma_f64', 'v_rcp_f64', 'v_xor_b32',
         'v_alignbit_b32', 'v_fma_f64 dependent chain', 'v_fma_f32 dependent chain', 'v_mul_f64', 'v_add_f64', 'v_cndmask_b32']
iters = 1000
def t(mode, threads, it):
    f = lambda: lib.run(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(inp.data_ptr()), it, 256, threads, None)
    assert f() == 0; torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3
ref = None
for threads in (256, 512):
    for mode in range(15):
        us = t(mode, threads, iters) - t(mode, threads, 0)
        ns = us * 1e3 / (iters * 64)
        if mode == 8 and threads == 256: ref = ns
        print(f'waves/SIMD={threads//256} {names[mode]:28s}: {ns:6.2f} ns/inst' + (f'  = {4*ns/ref:5.1f} clk if v_xor_b32 is 4' if ref else ''))
