"""What a device-to-device copy gets on this GPU: torch's copy_ (a vectorised elementwise kernel), and hipMemcpyAsync
through the C-ABI's runtime (the blit kernel).  K6 moves 8 B in + 8 B out per tuple, i.e. is a copy with a detour."""
import ctypes, os, sys
import torch
torch.cuda.init()
n = 2 << 30                       # 8 GiB of int32
a = torch.empty(n, dtype=torch.int32, device="cuda").fill_(3)
b = torch.empty_like(a)
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
ms = timed(lambda: b.copy_(a))
print("torch copy_ 8 GiB: %.3f ms = %.0f GB/s read + written" % (ms, 2 * 4 * n / ms / 1e6))
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
ms = timed(lambda: hip.hipMemcpyAsync(b.data_ptr(), a.data_ptr(), 4 * n, 3, st))
print("hipMemcpyAsync D2D 8 GiB: %.3f ms = %.0f GB/s read + written" % (ms, 2 * 4 * n / ms / 1e6))
ms = timed(lambda: b.fill_(7))
print("torch fill_ 8 GiB: %.3f ms = %.0f GB/s written" % (ms, 4 * n / ms / 1e6))
ms = timed(lambda: a.sum())
print("torch sum 8 GiB: %.3f ms = %.0f GB/s read" % (ms, 4 * n / ms / 1e6))
