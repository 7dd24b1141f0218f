python - <<'PY'
import time, ctypes
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
def t(f):
    t0=time.perf_counter(); r=f(); return time.perf_counter()-t0, r
hip.hipSetDevice(0); hip.hipFree(None)
for gb in (17, 24, 30, 33, 36, 48, 58, 120, 188):
    p=ctypes.c_void_p(); n=ctypes.c_size_t(gb*10**9)
    dt1,r1=t(lambda: hip.hipMalloc(ctypes.byref(p), n))
    dt2,r2=t(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
    dt4,_=t(lambda: hip.hipFree(p))
    print("%3d GB: hipMalloc %.3f s (rc %d), memset+sync %.3f s, hipFree %.3f s" % (gb, dt1, r1, dt2, dt4), flush=True)
# several 16 GB blocks alive at once
ps=[]; t0=time.perf_counter()
for i in range(11):
    p=ctypes.c_void_p(); r=hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(16*10**9)); ps.append(p)
print("11 x 16 GB hipMalloc alive together: %.3f s" % (time.perf_counter()-t0))
for p in ps: hip.hipFree(p)
PY
