"""What an allocation costs on this driver depending on the history of the memory it gets: 8 x 12 GB
from untouched VRAM, right after the same 96 GB were freed (sequentially and from 8 threads), 4 s after the free, and again
right after freeing memory that was never written.  Measured: 0.24 s / 3.98 s / 3.26 s / 0.00 s / 3.26 s — freed VRAM is
wiped in the background at ~30 GB/s whatever it held, and an allocation that lands on it waits for the wipe."""
import ctypes, time, threading
hip = ctypes.CDLL("libamdhip64.so")
def malloc(nbytes):
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
    assert rc == 0, rc
    return p
def free(p):
    hip.hipFree(p)
GB = 1 << 30
hip.hipSetDevice(0)
def seq(n, sz):
    t = time.time(); ps = [malloc(sz) for _ in range(n)]; hip.hipDeviceSynchronize(); dt = time.time() - t
    return ps, dt
def par(n, sz):
    ps = [None] * n
    def w(i): ps[i] = malloc(sz)
    t = time.time(); th = [threading.Thread(target=w, args=(i,)) for i in range(n)]
    [x.start() for x in th]; [x.join() for x in th]; hip.hipDeviceSynchronize(); dt = time.time() - t
    return ps, dt
# 1. fresh memory
ps, dt = seq(8, 12 * GB); print("fresh, sequential 8 x 12 GB: %.2f s" % dt, flush=True)
# touch it (memset) so it is dirty, then free
for p in ps: hip.hipMemset(p, 1, ctypes.c_size_t(12 * GB))
hip.hipDeviceSynchronize()
for p in ps: free(p)
ps, dt = seq(8, 12 * GB); print("dirty, sequential 8 x 12 GB: %.2f s" % dt, flush=True)
for p in ps: hip.hipMemset(p, 1, ctypes.c_size_t(12 * GB))
hip.hipDeviceSynchronize()
for p in ps: free(p)
ps, dt = par(8, 12 * GB); print("dirty, 8 threads x 12 GB: %.2f s" % dt, flush=True)
for p in ps: hip.hipMemset(p, 1, ctypes.c_size_t(12 * GB))
hip.hipDeviceSynchronize()
for p in ps: free(p)
time.sleep(4)
ps, dt = seq(8, 12 * GB); print("dirty, 4 s after the free, sequential: %.2f s" % dt, flush=True)
for p in ps: free(p)
ps, dt = seq(8, 12 * GB); print("freed untouched, sequential again: %.2f s" % dt, flush=True)
