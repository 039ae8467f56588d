#!/usr/bin/env python3
"""Soak of the streaming host driver: crawls with random worker counts, batch sizes, byte budgets and file sets for a
given time; every crawl of the same file set must report the same files / failed / frames / bytes.  A watchdog ends the
process (exit code 3) when one crawl takes longer than a minute.
usage: crawl_soak.py [seconds] [disk|db]   (disk: the pool is written to a temporary directory and the crawler reads the files
itself; db: every crawl writes the descriptor database and a digest of all its rows must repeat for the same file set)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from afec_amd import hostlib  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
from_disk = len(sys.argv) > 2 and sys.argv[2] == "disk"
with_db = len(sys.argv) > 2 and sys.argv[2] == "db"
rng = np.random.default_rng(2026)
base = bench.make_c3_files(48, 5)
pool = []
for i, x in enumerate(base):
    n = int(rng.integers(3000, len(x)))
    stereo = i % 3 == 0
    pcm = np.stack([x[:n], x[:n] // 2], axis=1).reshape(-1) if stereo else x[:n]
    # every fifth file at another sampling rate: converted on the GPU in front of LoadSample
    rate = [48000, 22050, 96000, 32000][(i // 5) % 4] if i % 5 == 1 else 44100
    pool.append(bench.wav_image(pcm, 2 if stereo else 1, rate))
pool.append(b"RIFF....not a wave file" * 4)
paths = None
if from_disk:
    import atexit
    import shutil
    import tempfile
    root = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    atexit.register(shutil.rmtree, root, True)
    paths = []
    for i, image in enumerate(pool):
        paths.append(os.path.join(root, f"pool{i:03d}.wav"))
        with open(paths[-1], "wb") as f:
            f.write(image)
    paths.append(os.path.join(root, "missing.wav"))          # a path that does not exist: a failed sample
    pool.append(None)
deadline = [time.time() + 60.0]


def watchdog():
    while True:
        time.sleep(1.0)
        if time.time() > deadline[0]:
            print("WATCHDOG: a crawl hangs", flush=True)
            os._exit(3)


threading.Thread(target=watchdog, daemon=True).start()
expected = {}
t_end = time.time() + seconds
crawls = files = 0
while time.time() < t_end:
    n = int(rng.choice([1, 7, 50, 300, 2000, 6000]))
    first = int(rng.integers(0, len(pool)))
    images = [pool[(first + i) % len(pool)] for i in range(n)]
    w = int(rng.integers(1, 13))
    b = int(rng.choice([1, 3, 16, 64, 256, 512, 2000]))
    hostlib.set_bytes_per_batch(int(rng.choice([0, 0, 1, 300000, 5000000])))
    if rng.random() < 0.05:
        hostlib.release()
    deadline[0] = time.time() + 60.0
    digest = None
    if with_db:
        import hashlib
        import sqlite3
        import tempfile
        n = min(n, 2000)
        images = images[:n]
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            st = hostlib.crawl(images, [f"f{i:05d}.wav" for i in range(n)], workers=w, files_per_batch=b, database=os.path.join(td, "soak.db"))
            con = sqlite3.connect(os.path.join(td, "soak.db"))
            h = hashlib.sha256()
            rows = 0
            for r in con.execute("SELECT * FROM assets ORDER BY filename"):
                rows += 1
                for k, v in enumerate(r):
                    if k != 1:                                   # modtime
                        h.update(v if isinstance(v, bytes) else repr(v).encode())
            con.close()
            assert rows == n, (rows, n)
            digest = h.hexdigest()
    elif from_disk:
        st = hostlib.crawl(None, [paths[(first + i) % len(pool)] for i in range(n)], workers=w, files_per_batch=b)
    else:
        st = hostlib.crawl(images, workers=w, files_per_batch=b)
    key = (n, first)
    sig = tuple(st[k] for k in ("files", "failed", "frames", "pcm_bytes", "result_bytes")) + (digest,)
    if expected.setdefault(key, sig) != sig:
        print("MISMATCH", key, expected[key], sig, (w, b), flush=True)
        sys.exit(2)
    crawls += 1
    files += n
hostlib.set_bytes_per_batch(0)
print(f"crawl soak{' (files on disk)' if from_disk else (' (database on, row digests compared)' if with_db else '')}: {crawls} crawls, {files} files in {seconds:.0f} s, no mismatch, no hang")
