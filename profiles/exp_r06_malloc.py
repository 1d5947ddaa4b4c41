"""How long do hipMalloc / hipFree of band-sized buffers take on this box (first-call cost of a fresh context)?"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
def t(f):
    t0 = time.perf_counter(); r = f(); hip.hipDeviceSynchronize(); return (time.perf_counter() - t0)*1e3, r
p0 = C.c_void_p(); hip.hipMalloc(C.byref(p0), 1 << 20); hip.hipDeviceSynchronize()
for gb in (0.25, 1, 4, 8):
    n = int(gb*(1 << 30))
    for rep in range(3):
        p = C.c_void_p()
        ms_a, rc = t(lambda: hip.hipMalloc(C.byref(p), n))
        ms_s, _ = t(lambda: hip.hipMemset(p, 0, n))
        ms_s2, _ = t(lambda: hip.hipMemset(p, 0, n))
        ms_f, _ = t(lambda: hip.hipFree(p))
        print("%.2f GB rep %d: hipMalloc %.2f ms (rc %d), first memset %.2f ms, second memset %.2f ms, hipFree %.2f ms" % (gb, rep, ms_a, rc, ms_s, ms_s2, ms_f))
