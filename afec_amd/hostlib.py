"""ctypes access to the C entry points of afec_amd/lib/libafx_host.so, the C++ host layer above the C-ABI
(afec_amd/host: WAV reader, streaming sharded crawler, sqlite descriptor pool)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repository root
_lib = None


def _bind(L):
    """argument types of the entry points whose pointers ctypes would otherwise truncate"""
    L.afec_wave_probe.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                  ctypes.c_int64, ctypes.c_char_p, ctypes.c_int32]
    L.afec_wave_probe_file.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_char_p, ctypes.c_int32]
    L.afec_shard_of_file.argtypes = [ctypes.c_int64, ctypes.c_int32]
    L.afec_crawl_wave_images.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p,
                                         ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int32]
    L.afec_crawl_wave_images_ex.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p,
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int32]
    return L


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "afec_amd", "csrc")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "afec_amd", "host")], stdout=subprocess.DEVNULL)
        _lib = _bind(ctypes.CDLL(os.path.join(ROOT, "afec_amd", "lib", "libafx_host.so")))
    return _lib


def wave_probe(image):
    """-> (dict of properties, decoded payload bytes) or raises RuntimeError with the reader's message."""
    L = lib()
    buf = np.frombuffer(image, dtype=np.uint8)
    props = (ctypes.c_int64 * 7)()
    err = ctypes.create_string_buffer(256)
    payload = np.zeros(2 * len(image) + 16, dtype=np.uint8)
    rc = L.afec_wave_probe(buf.ctypes.data, len(image), props, payload.ctypes.data, payload.size, err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    keys = ["channels", "rate", "bits", "sample_type", "frames", "raw_format", "payload_bytes"]
    d = dict(zip(keys, [int(v) for v in props]))
    return d, payload[:d["payload_bytes"]].tobytes()


def wave_probe_file(path):
    """wave_probe for a file on disk (the reader parses the head and reads the data chunk by pread)."""
    L = lib()
    props = (ctypes.c_int64 * 7)()
    err = ctypes.create_string_buffer(256)
    payload = np.zeros(2 * (os.path.getsize(path) if os.path.isfile(path) else 0) + 16, dtype=np.uint8)
    rc = L.afec_wave_probe_file(str(path).encode(), props, payload.ctypes.data, payload.size, err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    keys = ["channels", "rate", "bits", "sample_type", "frames", "raw_format", "payload_bytes"]
    d = dict(zip(keys, [int(v) for v in props]))
    return d, payload[:d["payload_bytes"]].tobytes()


def crawl(images, names=None, devices=(0,), workers=None, files_per_batch=512, database=None, digests=False):
    """images: list of bytes (WAV file images), or None: names are the paths of files on disk.  -> dict of statistics.
    workers: host threads per device; None = the library's choice (TCrawlOptions::mWorkersPerDevice = 0: from the CPUs the
    process may use and the number of devices, at most 5).  digests: also "row_digests", one uint64 per file over
    everything the device returned for it (0: not analysed)."""
    L = lib()
    n = len(images) if images is not None else len(names)
    names = names or [f"file{i:06d}.wav" for i in range(n)]
    c_names = (ctypes.c_char_p * n)(*[s.encode() for s in names])
    if images is not None:
        keep = [np.frombuffer(b, dtype=np.uint8) for b in images]
        c_images = (ctypes.c_void_p * n)(*[k.ctypes.data for k in keep])
        c_sizes = (ctypes.c_int64 * n)(*[len(b) for b in images])
    else:
        c_images = c_sizes = None
    G = len(devices)
    c_dev = (ctypes.c_int32 * G)(*devices)
    stats = (ctypes.c_double * (12 + G))()
    device_stats = (ctypes.c_double * (3 * G))()
    facts = (ctypes.c_double * 3)()
    row_digests = np.zeros(n, dtype=np.uint64) if digests else None
    err = ctypes.create_string_buffer(512)
    rc = L.afec_crawl_wave_images_ex(c_names, c_images, c_sizes, n, c_dev, G, int(workers or 0), files_per_batch,
                                     database.encode() if database else None, stats, device_stats,
                                     row_digests.ctypes.data if digests else None, facts, err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    keys = ["files", "failed", "frames", "pcm_bytes", "result_bytes", "seconds", "writer_seconds", "batches"]
    out = dict(zip(keys, list(stats)[:8]))
    out["files_per_device"] = [int(v) for v in list(stats)[8:8 + G]]
    out["cpu_seconds"] = stats[8 + G]   # process CPU time during the crawl: / seconds = busy CPUs
    out["skipped_sample_rate"] = int(stats[9 + G])   # files at another rate than the analyser's (not resampled here)
    out["retried_batches"] = int(stats[10 + G])      # GPU round trips that failed on a live device and were retried
    out["device_failed_files"] = int(stats[11 + G])  # files recorded as failed after their own attempts failed
    out["pcm_bytes_per_device"] = [int(device_stats[3 * d + 1]) for d in range(G)]
    out["seconds_per_device"] = [float(device_stats[3 * d + 2]) for d in range(G)]   # crawl start .. the device's last batch delivered
    out["workers_per_device"] = int(facts[0])
    out["usable_host_cpus"] = float(facts[1])
    out["aborted"] = bool(facts[2])          # request_abort() ended the crawl early: "files" counts what was analysed
    if digests:
        out["row_digests"] = row_digests
    return out


def fill_uniform_mt19937(n, seed):
    """n float32 U(-1, 1) from std::mt19937(seed) + std::uniform_real_distribution<float>(-1, 1): BASELINE configs[1]'s
    generator, the one the CPU-baseline driver (the reference's own objects) draws from."""
    L = lib()
    L.afec_fill_uniform_mt19937.restype = None
    L.afec_fill_uniform_mt19937.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32]
    x = np.empty(int(n), dtype=np.float32)
    L.afec_fill_uniform_mt19937(x.ctypes.data, int(n), int(seed) & 0xFFFFFFFF)
    return x


def usable_host_cpus():
    L = lib()
    L.afec_usable_host_cpus.restype = ctypes.c_double
    return float(L.afec_usable_host_cpus())


def workers_per_device_for(n_devices):
    L = lib()
    L.afec_workers_per_device_for.restype = ctypes.c_int32
    L.afec_workers_per_device_for.argtypes = [ctypes.c_int32]
    return int(L.afec_workers_per_device_for(int(n_devices)))


def request_abort():
    """Ends the crawl that is running in another thread of this process early (TCrawlOptions::mpAbortRequested -- the
    reference's SIGINT flag, Crawler.cpp:69-73, 717-720): no new batches, what was analysed is delivered."""
    L = lib()
    L.afec_crawl_request_abort.restype = None
    L.afec_crawl_request_abort()


def release():
    """Drop the crawlers the process keeps between crawl() calls (plans, device workspaces, page-locked buffers)."""
    L = lib()
    L.afec_crawl_release.restype = None
    L.afec_crawl_release()


def set_bytes_per_batch(n_bytes):
    """TCrawlOptions::mBytesPerBatch of the crawls that follow (0: the default, 128 MiB of file bytes)."""
    L = lib()
    L.afec_crawl_set_bytes_per_batch.restype = None
    L.afec_crawl_set_bytes_per_batch.argtypes = [ctypes.c_int64]
    L.afec_crawl_set_bytes_per_batch(int(n_bytes))


def set_database_pragmas(pragmas):
    """TCrawlOptions::mDatabasePragmas of the crawls that follow ("" or None: sqlite's defaults, like the reference)."""
    L = lib()
    L.afec_crawl_set_database_pragmas.restype = None
    L.afec_crawl_set_database_pragmas.argtypes = [ctypes.c_char_p]
    L.afec_crawl_set_database_pragmas(pragmas.encode() if pragmas else None)


def set_resample(on):
    """TCrawlOptions::mResample of the crawls that follow (False: files at another rate than the analyser's are skipped and counted)."""
    L = lib()
    L.afec_crawl_set_resample.restype = None
    L.afec_crawl_set_resample.argtypes = [ctypes.c_int32]
    L.afec_crawl_set_resample(1 if on else 0)


def set_test_fault(batch=-1, attempts=0, device_lost=False):
    """Fault injection (TCrawlOptions::mTestFailBatch / mTestFailAttempts / mTestDeviceLost) of the crawls that follow: the
    first `attempts` GPU attempts on files of the batch-th batch throw (-1: every attempt); defaults: no fault."""
    L = lib()
    L.afec_crawl_set_test_fault.restype = None
    L.afec_crawl_set_test_fault.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.afec_crawl_set_test_fault(int(batch), int(attempts), 1 if device_lost else 0)


def set_device_bytes_per_batch(n_bytes):
    """TCrawlOptions::mDeviceBytesPerBatch of the crawls that follow (0: the default, 2 GiB)."""
    L = lib()
    L.afec_crawl_set_device_bytes_per_batch.restype = None
    L.afec_crawl_set_device_bytes_per_batch.argtypes = [ctypes.c_int64]
    L.afec_crawl_set_device_bytes_per_batch(int(n_bytes))


def set_frame_kernel(frame_kernel):
    """TCrawlOptions::mFrameKernel of the crawls that follow: -1 the default (pinned: one frame per 64-lane wave), 0 the
    library's choice by batch size (not batch-independent), 1 / 2 the 64-lane / the half-wave layout."""
    L = lib()
    L.afec_crawl_set_frame_kernel.restype = None
    L.afec_crawl_set_frame_kernel.argtypes = [ctypes.c_int32]
    L.afec_crawl_set_frame_kernel(int(frame_kernel))
