#!/usr/bin/env python3
"""Page-locking rate on the GPU box: hipHostMalloc vs hipHostRegister of an anonymous mapping with and without
transparent huge pages (the first crawl of a process page-locks ~1.2 GB)."""
import ctypes
import mmap
import time

hip = ctypes.CDLL("libamdhip64.so")
libc = ctypes.CDLL(None, use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
libc.memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipHostFree.argtypes = [ctypes.c_void_p]
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
p = ctypes.c_void_p()
hip.hipHostMalloc(ctypes.byref(p), 1 << 20, 0); hip.hipHostFree(p)     # runtime init
N = 128 << 20
for rep in range(2):
    t = time.time(); hip.hipHostMalloc(ctypes.byref(p), N, 0); dt = time.time() - t
    print(f"hipHostMalloc 128 MiB: {dt * 1e3:.1f} ms"); hip.hipHostFree(p)
    for huge in (0, 1):
        t = time.time()
        a = libc.mmap(None, N + (2 << 20), mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
        b = (a + (2 << 20) - 1) & ~((2 << 20) - 1)
        if huge:
            libc.madvise(b, N, 14)     # MADV_HUGEPAGE
        libc.memset(b, 0, N)
        t1 = time.time()
        rc = hip.hipHostRegister(b, N, 0)
        t2 = time.time()
        print(f"mmap{' + MADV_HUGEPAGE' if huge else ''} + touch {1e3 * (t1 - t):.1f} ms, hipHostRegister {1e3 * (t2 - t1):.1f} ms (rc {rc})")
        hip.hipHostUnregister(b)
