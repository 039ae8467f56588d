import ctypes, time, threading
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipFree(p)
for mb in (1, 16, 128, 512):
    ps = []
    t = time.time()
    for _ in range(8):
        q = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(q), mb << 20); ps.append(q)
    dt = time.time() - t
    t = time.time()
    for q in ps: hip.hipFree(q)
    df = time.time() - t
    print(f"hipMalloc {mb} MiB: {dt / 8 * 1e3:.2f} ms each, hipFree {df / 8 * 1e3:.2f} ms each")
def work():
    for _ in range(4):
        q = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(q), 128 << 20)
t = time.time(); th = [threading.Thread(target=work) for _ in range(8)]; [x.start() for x in th]; [x.join() for x in th]
print(f"8 threads x 4 x 128 MiB: {1e3 * (time.time() - t):.1f} ms")
