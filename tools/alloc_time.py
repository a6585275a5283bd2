"""What hipMalloc / hipHostMalloc / hipFree cost on this box (hipMalloc of 18 GiB: nothing measurable -- the work is done when the
memory is first touched --; page-locking host memory: 0.14 s a GiB, releasing it 0.08 s): python tools/alloc_time.py"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def t_malloc(n):
    p = ctypes.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
    dt = time.perf_counter() - t
    return rc, dt, p
hip.hipSetDevice(0)
hip.hipFree(None)
for n in (1 << 30, int(3.5 * 2**30), 18 << 30, int(3.5 * 2**30), 1 << 30):
    rc, dt, p = t_malloc(n)
    t = time.perf_counter(); hip.hipMemset(p, 0, ctypes.c_size_t(min(n, 1 << 20))); hip.hipDeviceSynchronize(); d2 = time.perf_counter() - t
    t = time.perf_counter(); hip.hipFree(p); d3 = time.perf_counter() - t
    print("hipMalloc %.2f GiB: rc %d %.3f s; first touch %.4f s; hipFree %.3f s" % (n / 2**30, rc, dt, d2, d3))
hp = ctypes.c_void_p()
for n in (256 << 20, 512 << 20):
    t = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(hp), ctypes.c_size_t(n), 0); dt = time.perf_counter() - t
    t = time.perf_counter(); hip.hipHostFree(hp); d3 = time.perf_counter() - t
    print("hipHostMalloc %d MiB: rc %d %.3f s; free %.3f s" % (n >> 20, rc, dt, d3))
