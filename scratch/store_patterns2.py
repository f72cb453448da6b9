import ctypes, os, torch
lib = ctypes.CDLL(os.path.abspath('scratch/libstore2.so'))
R = 32 * 10240
buf = torch.empty(R * 2048, device='cuda', dtype=torch.uint8)
def t(which, N=2000, rpb=64, threads=128, stride=None, W=8, reps=20):
    stride = N if stride is None else stride
    f = lambda: lib.run(which, ctypes.c_void_p(buf.data_ptr()), R, N, rpb, threads, ctypes.c_size_t(stride), W, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert f() == 0; torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms * 1e3, R * N / ms / 1e9
print('rows N=2000 rpb=64 th=128          : %.1f us  %.2f TB/s' % t(1))
print('rows N=2048 (full aligned rows)    : %.1f us  %.2f TB/s' % t(1, N=2048))
print('rows N=1024 th=64 (one wave/row)   : %.1f us  %.2f TB/s' % t(1, N=1024, threads=64))
print('block-contig th=128 rpb=64         : %.1f us  %.2f TB/s' % t(2, threads=128))
print('block-contig th=256 rpb=64         : %.1f us  %.2f TB/s' % t(2, threads=256))
for W in (8, 16, 64):
    for th in (64, 256):
        print(f'wave-window W={W} th={th} rpw=64      : %.1f us  %.2f TB/s' % t(3, threads=th, W=W))
for th in (64, 256):
    print(f'wave-window W=8 th={th} rpw=16      : %.1f us  %.2f TB/s' % t(3, rpb=16, threads=th, W=8))
for W in (8, 16):
    for th in (128, 256):
        print(f'wave-window LDS W={W} th={th} rpw=64  : %.1f us  %.2f TB/s' % t(4, threads=th, W=W))
