#!/usr/bin/env python3
"""What hipMalloc costs on this pool after memory has been given back (raw HIP runtime through ctypes, no library): the driver wipes released
device memory in the background and the next allocation waits for it -- whether the memory was released by this process or by the process
that ran before.  One mode per process, run one after another (profiles/r04/alloc_after_free.txt):

    for m in no_free kernel_first free_small_wait free_small no_free; do python tools/alloc_after_free.py $m; done"""
import ctypes as C, sys, time, threading
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
def alloc(total, chunk):
    out = []
    left = total
    t0 = time.perf_counter()
    while left > 0:
        p = C.c_void_p(); b = min(chunk, left)
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(b))
        if rc: print("rc", rc); break
        out.append(p); left -= b
    return out, time.perf_counter() - t0
def free(o):
    t0 = time.perf_counter()
    for p in o: hip.hipFree(p)
    return time.perf_counter() - t0
mode = sys.argv[1]
f = C.c_size_t(); t = C.c_size_t()
hip.hipMemGetInfo(C.byref(f), C.byref(t)); print(mode, "free %.1f GB" % (f.value / 1e9))
if mode == "no_free":
    o, ta = alloc(16 << 20, 16 << 20)
    print("16 MiB: alloc %.4f (kept)" % ta)
    o2, ta = alloc(int(95e9), 1 << 30); print("95 GB, nothing freed before: %.3f s" % ta)
    hip.hipMemsetAsync(o[0], 1, C.c_size_t(16 << 20), None); hip.hipDeviceSynchronize()
    o3, ta = alloc(int(95e9), 1 << 30); print("95 GB more, after a memset kernel ran: %.3f s" % ta)
    for p in o3: hip.hipMemsetAsync(p, 1, C.c_size_t(1 << 30), None)
    t0 = time.perf_counter(); hip.hipDeviceSynchronize(); print("memset of those 95 GB: %.3f s" % (time.perf_counter() - t0))
elif mode == "kernel_first":
    o, ta = alloc(16 << 20, 16 << 20)
    hip.hipMemsetAsync(o[0], 1, C.c_size_t(16 << 20), None); hip.hipDeviceSynchronize()
    o2, ta = alloc(int(190e9), 1 << 30); print("190 GB after a kernel ran, nothing freed: %.3f s" % ta)
elif mode == "free_small_wait":
    o, ta = alloc(16 << 20, 16 << 20); tf = free(o)
    time.sleep(2.0)
    o2, ta = alloc(int(190e9), 1 << 30); print("190 GB, 2 s after a 16 MiB free: %.3f s" % ta)
elif mode == "free_small":
    o, ta = alloc(16 << 20, 16 << 20); tf = free(o)
    o2, ta = alloc(int(190e9), 1 << 30); print("190 GB right after a 16 MiB free: %.3f s" % ta)
