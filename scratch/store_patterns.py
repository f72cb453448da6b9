import ctypes, os, torch
lib = ctypes.CDLL(os.path.abspath('scratch/libstore.so'))
R, N = 32 * 10240, 2000
buf = torch.empty(R * 2048, device='cuda', dtype=torch.uint8)
def t(which, rpb=64, threads=128, stride=N, reps=20):
    f = lambda: lib.run(which, ctypes.c_void_p(buf.data_ptr()), R, N, rpb, threads, ctypes.c_size_t(stride), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms * 1e3, R * N / ms / 1e9
print('flat memset-like           : %.1f us  %.2f TB/s' % t(0))
for rpb in (16, 64, 256):
    for th in (128, 256):
        print(f'rows (K4 order) rpb={rpb} threads={th}: %.1f us  %.2f TB/s' % t(1, rpb, th))
print('rows, stride padded to 2048: %.1f us  %.2f TB/s' % t(1, 64, 128, 2048))
for rpb in (16, 64):
    print(f'block-contiguous rpb={rpb}     : %.1f us  %.2f TB/s' % t(2, rpb, 256))
