python - <<'PY'
import time, ctypes, threading
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0); hip.hipFree(None)
def malloc(gb):
    p=ctypes.c_void_p(); r=hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb*10**9))); return p, r
# dirty (almost) all of the device once
t0=time.perf_counter(); p,r=malloc(270); hip.hipMemset(p,1,ctypes.c_size_t(270*10**9)); hip.hipDeviceSynchronize(); hip.hipFree(p); print("dirtied 270 GB (rc %d) in %.2f s"%(r,time.perf_counter()-t0), flush=True)
def seq(k, gb):
    t0=time.perf_counter(); ps=[malloc(gb)[0] for _ in range(k)]; dt=time.perf_counter()-t0
    for p in ps: hip.hipFree(p)
    return dt
def par(k, gb):
    ps=[None]*k
    def w(i):
        hip.hipSetDevice(0); ps[i]=malloc(gb)[0]
    t0=time.perf_counter(); th=[threading.Thread(target=w,args=(i,)) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]; dt=time.perf_counter()-t0
    for p in ps: hip.hipFree(p)
    return dt
print("sequential 8 x 16 GB: %.2f s" % seq(8,16), flush=True)
print("8 threads x 16 GB:    %.2f s" % par(8,16), flush=True)
print("sequential 8 x 16 GB: %.2f s" % seq(8,16), flush=True)
print("16 threads x 8 GB:    %.2f s" % par(16,8), flush=True)
print("one 128 GB:           %.2f s" % seq(1,128), flush=True)
print("64 threads x 2 GB:    %.2f s" % par(64,2), flush=True)
PY
