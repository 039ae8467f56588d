import os, sys, time, ctypes, threading
sys.path.insert(0, os.getcwd())
import bench
from afec_amd import hostlib
files = bench.make_c4_files(64, 1234)
pool = [bench.wav_image(f, 2) for f in files]
images = [pool[i % len(pool)] for i in range(12500)]
L = hostlib.lib()
# page-locking rate, alone and from 8 threads
hip = ctypes.CDLL(os.path.join(hostlib.ROOT, "afec_amd", "lib", "libafx_hip.so"))
hip.afx_host_alloc.restype = ctypes.c_void_p; hip.afx_host_alloc.argtypes = [ctypes.c_int64]
hip.afx_host_free.argtypes = [ctypes.c_void_p]
t = time.time(); p = hip.afx_host_alloc(1 << 20); print("first 1 MiB (runtime init)", round((time.time() - t) * 1e3, 1), "ms"); hip.afx_host_free(p)
for mb in (32, 113):
    t = time.time(); p = hip.afx_host_alloc(mb << 20); dt = time.time() - t; print(f"{mb} MiB pinned: {dt * 1e3:.1f} ms = {mb / 1024 / dt:.1f} GiB/s"); 
    t = time.time(); hip.afx_host_free(p); print(f"   free {1e3 * (time.time() - t):.1f} ms")
def one(): 
    p = hip.afx_host_alloc(113 << 20); ps.append(p)
ps = []
t = time.time(); th = [threading.Thread(target=one) for _ in range(8)]; [x.start() for x in th]; [x.join() for x in th]
print(f"8 x 113 MiB from 8 threads: {1e3 * (time.time() - t):.1f} ms")
for p in ps: hip.afx_host_free(p)
for k in range(3):
    t = time.time(); st = hostlib.crawl(images, workers=8, files_per_batch=512); print(f"crawl {k}: {st['seconds'] * 1e3:.1f} ms (call {1e3 * (time.time() - t):.1f} ms)")
