#!/usr/bin/env python3
"""bench.py -- audio frames/s of the low-level spectral hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 2048/1024 STFT + 14-coefficient MFCC on synthetic
44.1 kHz mono float32 PCM, U(-1,1) from std::mt19937 + std::uniform_real_distribution<float>(-1, 1) -- SURVEY 8(d)'s
generator, the one the reference-side timing driver draws from; buffer b of rank r is seeded 1234 + 64 r + b, so buffer 0
of rank 0 is the survey's mt19937(1234) stream itself: `--buffers` buffers of exactly
10 000 frames each per GPU (512 by default: 5.12 M frames, 21 GB of PCM resident in HBM; 64 of them are generated,
larger batches tile those -- every copy has its own place in HBM).  One *step* = one pass of the HIP path over that
whole batch, PCM already resident in HBM.  The batch is sized so that a step takes ~10 ms: from an idle GPU the clocks
need ~30 ms of load to settle (the first 25 launches of a 1.3 ms step run 5-15 % slower, profiles/r03/clock_ramp.txt),
and a handful of warm-up steps must cover that.  N>1: one process per GPU, every rank owns its own batch
(files are sharded, no data-path collective) -> weak scaling; value = frames all ranks processed
/ max-over-ranks time.  The ranks come from an outside launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, run plainly as `python bench.py --gpus N`, from this script itself: before anything touches the
GPU it starts N children of itself with those variables set (launch_ranks), relays rank 0's line and fails when
any child does -- the crawler's own fan-out of one self-contained task per file (Crawler.cpp:706-728) one level up.
After the replica measurement ONE process (rank 0) drives all N devices through the C++ file-sharding crawler
(afec::TCrawler, file i -> device i mod N: the analogue of Crawler.cpp:706-728) over BASELINE configs[3] -- 12 500 x N
stereo one-second WAV images -- and the line carries that as config.sharded_crawl.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

T_SCRIPT_START = time.time()     # the line's "run_seconds" counts from here (the interpreter's own start-up is before it)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The end-to-end leg is a pipeline with eight batches in flight: it wants 16 hardware queues (afec::TCrawler asks for
# them itself, TCrawlOptions::mHardwareQueues, but the HIP runtime reads the variable at its first call, which in this
# script is the headline plan's).  The headline measurement uses one stream and does not depend on it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

import afec_amd as afx  # noqa: E402  (loads nothing GPU-side until a Plan is created)

FRAMES_PER_BUFFER = 10000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# what a streaming read kernel reaches on this part (tools/ubench/hbm_stream.hip, profiles/r02/ubench_hbm_stream.txt;
# a copy reaches 4 700-5 400): SURVEY 8(d) asks for this denominator beside the nominal one
HBM_ACHIEVABLE_READ_GBS = 6400.0
NOMINAL_CLOCK_GHZ = 2.4      # MI355X peak engine clock (MI355X_MICROARCH.md); under the f64 kernels the part sustains 1.9-2.1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--buffers", type=int, default=512, help="10k-frame buffers per GPU per step")
    ap.add_argument("--precision", choices=["f64"], default="f64", help="the arithmetic the path computes in (the reference's)")
    ap.add_argument("--mask", default="c2", choices=["c2", "star", "stats", "all", "frame", "neighbours", "everything"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4"],
                    help="c2: --buffers x 10k-frame buffers (headline); c3: 1000 synthetic 2.0 s files, full "
                         "low-level set + per-file statistics (BASELINE.json configs[2]); c4: this rank's share "
                         "of 100 000 stereo 1.0 s files (configs[3]: 12 500 per GPU on 8 GPUs, --files to change)")
    ap.add_argument("--files", type=int, default=12500, help="files per GPU of the c4 workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-10k-frame-buffer figure (profiling runs)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="with --workload c3|c4: time the streaming host driver instead (WAV images in host memory -> RIFF "
                         "parse -> page-locked staging -> upload -> LoadSample + every descriptor + statistics -> results "
                         "back in host memory); value = frames/s including every transfer")
    ap.add_argument("--workers", type=int, default=5, help="host threads (batches in flight) per GPU of --end-to-end")
    ap.add_argument("--files-per-batch", type=int, default=512, help="files per GPU batch of --end-to-end")
    ap.add_argument("--cpu-frames", type=int, default=30000, help="frames per CPU worker for the baseline")
    ap.add_argument("--frame-kernel", choices=["auto", "wave64", "halfwave"], default="auto",
                    help="afx_plan_desc.frame_kernel: layout of the STFT kernel (A/B timing; auto = by batch size)")
    ap.add_argument("--no-side-stream", action="store_true",
                    help="AFX_PLAN_NO_SIDE_STREAM: the rhythm tracker's kernels on the batch's own stream (profiles: a "
                         "kernel's duration is then its own)")
    ap.add_argument("--no-spot-check", action="store_true", help="skip the parity spot check of the timed batch")
    ap.add_argument("--batch-files", type=int, default=0,
                    help="with --workload c3|c4: cut the files into batches of this many (the crawler's shape: 512) instead of "
                         "one batch; a step is then one pass over all of them, --in-flight at a time on their own streams")
    ap.add_argument("--in-flight", type=int, default=5, help="batches in flight with --batch-files (TCrawler: one per worker)")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not sample the shader clock during the timed launches")
    ap.add_argument("--no-sharded-crawl", action="store_true", help="skip config.sharded_crawl (rank 0 driving all N devices)")
    ap.add_argument("--no-chain-rates", action="store_true",
                    help="skip the C3 / C4-share chain rates (BASELINE configs[2] / [3]) of the default line")
    # test hook of the launcher (tests/test_bench_dist_cpu.py): this rank exits with status 7 before it joins the others
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--stall-seconds", type=float, default=0.0, help=argparse.SUPPRESS)   # test hook: every rank sleeps first
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_gpus, argv, poll_s=0.05):
    """`python bench.py --gpus N` with no launcher around it: N child processes of this very script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- what torchrun would export.  Called before this
    process has touched the GPU (nothing is ever exec'ed over an initialised one: the children are new processes and
    this one only waits).  Rank 0's stdout is relayed; when a child fails the others are ended (by the PIDs started
    here) and the failing status is returned -- a rank waiting in a barrier for a dead peer must not hang the job.
    Returns the exit status for this process."""
    import ctypes
    import signal
    import tempfile
    port = _free_port()
    children = []

    # a launcher that is told to stop (the driver's timeout: SIGTERM) must not leave its ranks behind on the GPUs: the
    # handlers are in place before the first rank exists (a signal during start-up ends the ranks started so far), and
    # a launcher that is killed outright takes its ranks with it (PR_SET_PDEATHSIG in the child before it execs)
    def stop(signum, _frame):
        raise SystemExit(128 + signum)
    previous = {sig: signal.signal(sig, stop) for sig in (signal.SIGTERM, signal.SIGINT)}

    try:
        libc = ctypes.CDLL("libc.so.6")      # looked up here, not in the forked child
    except OSError:
        libc = None

    def die_with_launcher():
        if libc is not None:
            libc.prctl(1, int(signal.SIGTERM))      # PR_SET_PDEATHSIG
    status = 0
    pending = set()
    try:
        for r in range(n_gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), AFX_BENCH_LAUNCHED_BY=str(os.getpid()))
            out = tempfile.TemporaryFile()
            children.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, stdout=out,
                                              preexec_fn=die_with_launcher), out))
            pending.add(r)
        while pending and status == 0:
            for r in sorted(pending):
                rc = children[r][0].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    print(f"bench.py: rank {r} of {n_gpus} exited with status {rc}; ending the other ranks", file=sys.stderr)
                    status = rc if rc > 0 else 1
                    break
            if pending and status == 0:
                time.sleep(poll_s)
    finally:
        for r in pending:                  # after a failure or a signal: the ranks that are still running (by their PIDs)
            if children[r][0].poll() is None:
                children[r][0].terminate()
        for r in pending:
            try:
                children[r][0].wait(10)
            except subprocess.TimeoutExpired:
                children[r][0].kill()
                children[r][0].wait()
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    for r, (_, out) in enumerate(children):
        out.seek(0)
        text = out.read().decode(errors="replace")
        out.close()
        if r == 0 and status == 0:
            for line in text.splitlines():          # the line itself to stdout, anything a library printed to stderr
                (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
            sys.stdout.flush()
        elif text.strip():
            sys.stderr.write(f"[rank {r} stdout] {text}")
    return status


DISTINCT_BUFFERS = 64   # mt19937 buffers generated on the host; larger batches tile them (every copy has its own place in HBM)
C2_SEED = 1234          # std::mt19937(1234)


def mt19937_uniform(n, seed):
    """n float32 of std::uniform_real_distribution<float>(-1, 1) over std::mt19937(seed): the host library's
    helper (afec_fill_uniform_mt19937: the C++ standard library itself), else the same arithmetic restated with numpy's
    MT19937 under the classic init_genrand seeding -- libstdc++ draws one 32-bit word per float: float(word) / 2^32,
    a result of 1.0 replaced by the float below it, then x 2 - 1 in float (tests/test_host_wav_cpu.py pins the two on
    each other)."""
    try:
        from afec_amd import hostlib
        return hostlib.fill_uniform_mt19937(n, seed)
    except Exception as e:  # noqa: BLE001  (the headline must not depend on the host library)
        print(f"warning: afec_fill_uniform_mt19937 unavailable ({e}); numpy restatement of the same generator", file=sys.stderr)
        bg = np.random.MT19937()
        bg._legacy_seeding(int(seed) & 0xFFFFFFFF)
        u = bg.random_raw(n).astype(np.uint32).astype(np.float32) / np.float32(4294967296.0)
        u = np.where(u >= np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(0.0)), u).astype(np.float32)
        return (np.float32(2.0) * u + np.float32(-1.0)).astype(np.float32)


def make_buffers(n_buffers, seed):
    """buffer b: std::mt19937(seed + b) -> U(-1, 1) floats; with seed = C2_SEED buffer 0 is SURVEY 8(d)'s stream"""
    from concurrent.futures import ThreadPoolExecutor
    n = (FRAMES_PER_BUFFER - 1) * 1024 + 2048
    distinct = min(n_buffers, DISTINCT_BUFFERS)
    with ThreadPoolExecutor(max_workers=min(8, distinct)) as pool:      # the C++ helper runs without the interpreter lock
        bufs = list(pool.map(lambda b: mt19937_uniform(n, seed + b), range(distinct)))
    return [bufs[i % len(bufs)] for i in range(n_buffers)]


class ClockProbe:
    """tools/clock_probe: one wave on a stream of its own samples s_memtime / s_memrealtime every ~25 us while launches
    run; stop() -> {"clock_ghz", "min_ghz_1ms", "max_ghz_1ms", "samples", "seconds", "ended_by"} or None."""

    def __init__(self, device, max_seconds):
        import ctypes
        self.handle = None
        path = os.path.join(ROOT, "afec_amd", "lib", "libafx_clock_probe.so")
        if not os.path.exists(path):
            subprocess.call(["make", "-C", os.path.join(ROOT, "tools", "clock_probe")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            self.lib = ctypes.CDLL(path)
        except OSError as e:
            print(f"warning: clock probe not available ({e}): roofline.clock_ghz_in_run is null", file=sys.stderr)
            return
        self.lib.afx_clock_probe_start.restype = ctypes.c_void_p
        self.lib.afx_clock_probe_start.argtypes = [ctypes.c_int, ctypes.c_double]
        self.lib.afx_clock_probe_stop.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        self.handle = self.lib.afx_clock_probe_start(int(device), float(max_seconds))
        if not self.handle:
            print("warning: the clock probe did not start: roofline.clock_ghz_in_run is null", file=sys.stderr)

    def stop(self):
        import ctypes
        if not self.handle:
            return None
        out = (ctypes.c_double * 7)()
        rc = self.lib.afx_clock_probe_stop(self.handle, out)
        self.handle = None
        if rc != 0:
            return None
        return {"clock_ghz": out[0], "min_ghz_1ms": out[1], "max_ghz_1ms": out[2], "samples": int(out[3]), "seconds": out[4],
                "ended_by": {1: "stop", 2: "its own time limit"}.get(int(out[5]), "?"), "wall_clock_khz": out[6]}


def clock_during(run, device, expect_seconds):
    """The shader clock the device sustains under run() -- a REPEAT of launches that have just been timed, in the same
    process, seconds later -- sampled by the probe wave beside them.  Not during the timed region itself: the f64 frame
    kernels fill every SIMD's register file (two waves of 256 VGPRs), so the resident probe wave displaces one
    workgroup of the persistent grid and the pass with it runs ~4 % slower (measured: 490 against 512 M frames/s,
    profiles/r06/clock_probe_cost.txt); the load, and with it the clock, is the same but for that one workgroup in 256.
    -> (clock dict or None, milliseconds run() reported).  A probe that ran into its own time limit (its stream shared a
    hardware queue with the launches, which then waited for it) is reported as no clock."""
    probe = ClockProbe(device, min(20.0, 4.0 * expect_seconds + 0.5))
    ms = run()
    clock = probe.stop()
    if clock is not None and clock["ended_by"] != "stop":
        print("warning: the clock probe ran into its time limit (a shared hardware queue): no clock for this measurement", file=sys.stderr)
        clock = None
    if clock is not None:
        clock["how"] = ("one wave on its own stream, s_memtime / s_memrealtime every ~25 us (tools/clock_probe) during a repeat of the timed "
                        "launches right after them; the probe wave displaces one workgroup of the persistent grid, so it is not resident "
                        "during the timed region itself")
        clock["ms_per_pass_with_the_probe_resident"] = ms
    return clock, ms


def make_c3_files(n_files, seed):
    """2.0 s mono 16-bit PCM files (88 200 samples): 1-3 sines, a decaying noise burst,
    50 ms of leading silence.  They go through the LoadSample front end on the GPU
    (afx_batch_create_from_raw: normalisation, -48 dB trim, padding), outside the timed region."""
    rng = np.random.Generator(np.random.MT19937(seed))
    n = 88200
    t = np.arange(n, dtype=np.float64) / 44100.0
    files = []
    for _ in range(n_files):
        x = np.zeros(n)
        for _ in range(int(rng.integers(1, 4))):
            x += rng.uniform(0.2, 0.6) * np.sin(2 * np.pi * rng.uniform(110.0, 4000.0) * t + rng.uniform(0, 6.28))
        x += rng.uniform(0.2, 0.8) * rng.uniform(-1, 1, n) * np.exp(-t / rng.uniform(0.05, 0.5))
        x[:2205] = 0.0
        x *= rng.uniform(0.3, 0.95) / np.max(np.abs(x))
        files.append(np.round(x * 32767.0).astype(np.int16))
    return files


def make_c4_files(n_files, seed):
    """stereo 16-bit 1.0 s files, right = left delayed by 7 samples x 0.8.  A pool of 64 distinct
    files is cycled (the kernels are data-independent in time; generating 12 500 distinct files takes minutes)."""
    pool = []
    for x in make_c3_files(64, seed):
        left = x[:44100].astype(np.float64)
        right = 0.8 * np.concatenate([np.zeros(7), left[:-7]])
        pool.append(np.stack([left, right], axis=1).round().astype(np.int16).reshape(-1))
    return [pool[i % len(pool)] for i in range(n_files)]


def wav_image(pcm_i16, channels, rate=44100):
    """a 16-bit PCM RIFF / WAVE file image"""
    import struct
    payload = np.ascontiguousarray(pcm_i16).tobytes()
    fmt = struct.pack("<HHIIHH", 1, channels, rate, channels * rate * 2, channels * 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _c4_images(n_files, seed, distinct=64):
    pool = [wav_image(f, 2) for f in make_c4_files(64, seed)[:distinct]]
    return [pool[i % len(pool)] for i in range(n_files)]


DISTINCT_CRAWL_CONTENTS = 63    # odd: with files sharded i mod G (G = 2, 4, 8) every content visits every device


def sharded_crawl(n_gpus, files_per_gpu, seed, pinned_device=None, repeats=4):
    """BASELINE configs[3] through ONE process and the C++ file-sharding driver: files_per_gpu x n_gpus stereo one-second
    WAV images, file i -> device i mod n_gpus (afec::TCrawler over all devices: the analogue of the reference's pool of
    one self-contained task per file, Crawler.cpp:706-728, its one writer SampleAnalyser.cpp:413-415), workers per
    device from the CPU quota.  pinned_device: every shard on that device (AFX_BENCH_DEVICE, 1-GPU boxes)."""
    from afec_amd import hostlib
    visible = afx.device_count()
    if pinned_device is not None:
        devices = [pinned_device] * n_gpus
    elif visible >= n_gpus:
        devices = list(range(n_gpus))
    else:
        devices = [i % max(1, visible) for i in range(n_gpus)]
    n_files = files_per_gpu * n_gpus
    images = _c4_images(n_files, seed, DISTINCT_CRAWL_CONTENTS)
    best, cold, busy = None, None, []
    for _ in range(repeats):
        st = hostlib.crawl(images, devices=devices, workers=None, files_per_batch=512)
        if cold is None:
            cold = st["seconds"]
        else:
            busy.append(st["cpu_seconds"] / st["seconds"])
        if best is None or st["seconds"] < best["seconds"]:
            best = st
    # the same content must give the same row whichever device / batch it landed on: digests over everything the device
    # returned per file, on a crawl in which every content visits every device twice
    check = hostlib.crawl(images[:2 * DISTINCT_CRAWL_CONTENTS * n_gpus], devices=devices, workers=None, files_per_batch=512, digests=True)
    dig = check["row_digests"]
    by_content = [set(int(v) for v in dig[c::DISTINCT_CRAWL_CONTENTS]) for c in range(DISTINCT_CRAWL_CONTENTS)]
    identical = bool(np.all(dig != 0)) and all(len(v) == 1 for v in by_content)
    return {"workload": f"C4: {n_files} stereo 1.0 s 16-bit WAV images in host memory ({DISTINCT_CRAWL_CONTENTS} distinct contents cycled), "
                        f"one process, afec::TCrawler over {n_gpus} shards, file i -> shard i mod {n_gpus}; LoadSample + every "
                        f"low-level descriptor (per-frame set + rhythm tracker) + statistics -> records back in host memory; "
                        f"every transfer inside the timed region; best of {repeats - 1} warm crawls",
            "devices": devices, "devices_visible": visible,
            "files": int(best["files"]), "failed": int(best["failed"]), "frames": int(best["frames"]),
            "files_per_s": best["files"] / best["seconds"], "frames_per_s": best["frames"] / best["seconds"],
            "seconds": best["seconds"], "cold_files_per_s": best["files"] / cold,
            "files_per_device": best["files_per_device"],
            "upload_GB_per_s_per_device": [b / s / 1e9 if s > 0 else None for b, s in zip(best["pcm_bytes_per_device"], best["seconds_per_device"])],
            "upload_GB_per_s": best["pcm_bytes"] / best["seconds"] / 1e9,
            "busy_host_cpus": float(np.median(busy)) if busy else best["cpu_seconds"] / best["seconds"],
            "cpu_quota": best["usable_host_cpus"], "workers_per_device": best["workers_per_device"], "files_per_batch": 512,
            "retried_batches": best["retried_batches"],
            "row_digests": {"files": int(dig.size), "contents": DISTINCT_CRAWL_CONTENTS, "rows_per_content": int(dig.size // DISTINCT_CRAWL_CONTENTS),
                            "files_per_device": check["files_per_device"], "identical_per_content": identical}}


def end_to_end(workload, n_files, device, workers, seed, database=None, repeats=3, files_per_batch=512, files_dir=None, rate=44100):
    """The streaming host driver (afec_amd/host/Crawler.cpp) on this rank's share of the crawl: files/s and frames/s
    with every transfer inside the timed region.  The crawler (plans, device workspaces, page-locked buffers) persists
    between the repeats: the first one is the cold crawl ("cold_seconds"), the best one the warm rate.
    files_dir: the files are written there first (untimed) and the crawler reads them from disk itself.
    rate: the sampling rate the files claim (another one than 44 100: converted on the GPU, SampleAnalyser.cpp:563-607)."""
    from afec_amd import hostlib
    files = make_c3_files(64, seed) if workload == "c3" else make_c4_files(64, seed)
    channels = 1 if workload == "c3" else 2
    pool = [wav_image(f, channels, rate) for f in files]
    images = [pool[i % len(pool)] for i in range(n_files)]
    names = None
    if files_dir is not None:
        names = [os.path.join(files_dir, f"{i // 1000:03d}", f"file{i:06d}.wav") for i in range(n_files)]
        for i in range(0, n_files, 1000):
            os.makedirs(os.path.dirname(names[i]), exist_ok=True)
        for name, image in zip(names, images):
            with open(name, "wb") as f:
                f.write(image)
        images = None
    best, cold, busy = None, None, []
    for rep in range(repeats):
        # every repeat writes its own database file: a crawl into the file of the one before would time INSERT OR REPLACE
        # of existing rows (delete + insert), not the inserts of a crawl
        db = f"{database}.{rep}" if database else None
        st = hostlib.crawl(images, names, devices=(device,), workers=workers, files_per_batch=files_per_batch, database=db)
        if cold is None:
            cold = st["seconds"]
        else:
            busy.append(st["cpu_seconds"] / st["seconds"])     # warm crawls only: the first one also sets the crawler up
        if best is None or st["seconds"] < best["seconds"]:
            best = st
    # busy host CPUs: the median over the warm crawls (the fastest crawl's own figure is not the typical one: a crawl is
    # fastest when its threads happened to wait least)
    best = dict(best, cold_seconds=cold,
                busy_cpus_median=float(np.median(busy)) if busy else best["cpu_seconds"] / best["seconds"])
    return best


def dist_setup(n_gpus):
    """Returns (rank, world, local_rank, dist or None).  N>1: the ranks were started by torchrun or by launch_ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, 1, 0, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # replicas only: the collective is a barrier and a max over ranks of one float on the host.
    # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): this script's stdout is ONE JSON
    # line, so file descriptor 1 points at stderr while the group forms.
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    return rank, world, local, dist


def reduce_max_sum(dist, seconds, frames):
    if dist is None:
        return seconds, frames
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    f = torch.tensor([float(frames)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), float(f.item())


def gather_ranks(dist, info):
    """every rank's {rank, device, ms_per_step, ...}, in rank order, on every rank"""
    if dist is None:
        return [info]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, info)
    return sorted(got, key=lambda d: d["rank"])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs the container may use at once (cgroup CPU bandwidth limit), None when unlimited / unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_baseline(frames_per_worker):
    """The reference CPU path timed on this host: oracle/_ref/ref_driver (the reference's own
    objects, kind "reference") when present, else the oracle restatement (kind "port").  One
    worker process per host core, the analogue of the crawler's one-file-per-task thread pool
    (Crawler.cpp:706-728); before that one process alone, for the un-contended single-thread rate."""
    host_cpus = os.cpu_count() or 1
    try:
        host_cpus_usable = len(os.sched_getaffinity(0))
    except AttributeError:
        host_cpus_usable = host_cpus
    # worker processes = the CPUs this job can really run on: the affinity mask, cut by the container's CPU bandwidth
    # quota (on the GPU pool: 256 hardware threads visible, quota 16 CPUs -- more processes than that only contend:
    # measured 544 k frames/s with 16 processes, 500 k with 64, 320 k with 256)
    quota = cpu_quota()
    cores = max(1, min(host_cpus_usable, 256, int(quota + 0.5) if quota else 256))
    host = {"cpu_model": cpu_model(), "os_cpu_count": host_cpus, "cpus_usable": host_cpus_usable, "cpu_quota": quota}
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(ref) and os.access(ref, os.X_OK):
        alone = subprocess.run([ref, "time", str(max(2000, frames_per_worker // 4)), "999"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL)
        alone_rate = json.loads(alone.stdout.decode())["frames_per_s"] if alone.returncode == 0 else None
        t0 = time.perf_counter()
        procs = [subprocess.Popen([ref, "time", str(frames_per_worker), str(1234 + i)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for i in range(cores)]
        outs = [p.communicate()[0] for p in procs]
        ok = all(p.returncode == 0 for p in procs)
        dt = time.perf_counter() - t0
        if ok:
            per = [json.loads(o.decode())["frames_per_s"] for o in outs]
            return dict(host, **{
                "value": cores * frames_per_worker / dt, "unit": "frames/s", "cores": cores, "kind": "reference",
                "sample": f"{cores} processes x {frames_per_worker} frames of U(-1,1), window+FFT+magnitude+xtract_mfcc "
                          f"via the reference's own objects (oracle/_ref/ref_driver time)",
                "single_thread_frames_per_s": alone_rate,
                "per_process_frames_per_s_while_all_run": float(np.median(per))})
    # port: the oracle restatement through ctypes in worker processes
    import multiprocessing as mp
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_oracle_worker, [(frames_per_worker, 1234 + i) for i in range(cores)])
    dt = time.perf_counter() - t0
    return dict(host, **{"value": cores * frames_per_worker / dt, "unit": "frames/s", "cores": cores, "kind": "port",
                         "sample": f"{cores} processes x {frames_per_worker} frames of U(-1,1), STFT+MFCC via oracle/afx_oracle.c"})


def _oracle_worker(arg):
    frames, seed = arg
    from tests._oracle import Oracle
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (frames - 1) * 1024 + 2048)
    return float(Oracle().run_mfcc(x)[0, 0])


def kernel_profile(precision, mask_name, workload, shape=None):
    """The committed rocprofv3 profile of this configuration (profiles/kernel_profiles.json, written from
    tools/profile_config.py runs): HBM bytes and VALU-pipe cycles per frame, the kernels of one step, and what the
    counters say limits it.  None when the configuration was not profiled."""
    p = os.path.join(ROOT, "profiles", "kernel_profiles.json")
    if not os.path.exists(p):
        return None
    try:
        if workload == "c2":
            key = f"{mask_name}_{precision}"
        else:   # c3 / c4: "c4" (per-frame set, one batch), "c4_everything", "c3_all" (BASELINE.md C3's spectral set), "c4_crawler" (the crawler's batch shape)
            key = workload + {"frame": "", "neighbours": "", "everything": "_everything", "all": "_all"}.get(mask_name, f"_{mask_name}") + (f"_{shape}" if shape else "")
        all_profiles = json.load(open(p))
        prof = all_profiles.get(key)
    except Exception:
        return None
    if prof is None:
        return None
    # counters belong to the build they were measured on: afx_build_info() carries a hash of the library's sources
    measured_on, loaded = all_profiles.get("_build_info"), afx.build_info()
    if measured_on != loaded:
        print(f"warning: profiles/kernel_profiles.json was measured on [{measured_on}], the loaded library is [{loaded}]: "
              f"roofline.traffic / roofline.valu are not quoted (re-run tools/profile_config.py + tools/make_kernel_profiles.py)",
              file=sys.stderr)
        return {"stale": True, "measured_on": measured_on}
    return prof


def parity_spot_check(batch, spot_bufs, frames_per_buffer, n_frames=64):
    """MFCC of the first n_frames frames of the given buffers, as the timed launches left them, against the oracle."""
    from tests import _tol
    from tests._oracle import Oracle
    oracle = Oracle()
    got = batch.fetch()["mfcc"]
    worst = over = 0.0
    rtol, atol = _tol.GPU_TOL["mfcc"]
    ok = True
    for i, x in spot_bufs.items():
        ref = oracle.run_mfcc(x[:2048 + 1024 * (n_frames - 1)].astype(np.float64))
        mine = got[i * frames_per_buffer:i * frames_per_buffer + n_frames]
        err = np.abs(mine - ref) / np.maximum(np.abs(ref), 1e-9)
        worst = max(worst, float(err.max()))
        over = max(over, _tol.over_ceiling("mfcc", mine, ref))
        ok = ok and bool(np.all(np.abs(mine - ref) <= atol + rtol * np.abs(ref)))
    return {"frames": n_frames * len(spot_bufs), "buffers": sorted(spot_bufs), "descriptor": "mfcc", "max_rel_err": worst,
            "rtol": rtol, "atol": atol, "worst_over_ceiling": over, "ceiling": _tol.OBSERVED_CEILING["mfcc"],
            "passed": bool(ok), "inside_ceiling": bool(over <= 1.0),   # the bar decides; an excess over the 10 x-observed ceiling on random material is reported, not fatal (tests/_tol.py)
            "against": "oracle/afx_oracle.c (pinned on the reference's objects)"}


SURVEY_C3_BYTES_PER_FRAME = 5080   # 4 096 B of PCM + 123 doubles of descriptors per frame of the full low-level set


def chain_spot_check(targets, channels, mask_name="frame"):
    """Every per-frame descriptor of the picked files, as the timed launches left them in HBM, against the oracle
    pipeline (load_sample -> run / run_neighbours): the bar of tests/_tol.py and the regression ceiling.
    targets: [(batch, index of the file inside that batch, its decoded PCM, its index in the workload)]."""
    from tests import _oracle, _tol
    from tests._oracle import FIELDS, NEIGH_FIELDS, Oracle
    ora = Oracle()
    fetched = {}
    worst, worst_field, over, ok, frames, values = 0.0, None, 0.0, True, 0, 0
    for batch, i, pcm, index in targets:
        if id(batch) not in fetched:
            fetched[id(batch)] = batch.fetch()
        res = fetched[id(batch)]
        off = res["frame_offset"]
        mono, _ = _oracle.load_sample(pcm, channels)
        ref = ora.run(mono, cap=True)
        if off[i + 1] - off[i] != ref.shape[0]:
            return {"passed": False, "error": f"file {index}: {off[i + 1] - off[i]} frames, the oracle has {ref.shape[0]}"}
        frames += ref.shape[0]
        pairs = [(f, res[f][off[i]:off[i + 1]].reshape(ref.shape[0], -1), ref[:, a:b]) for f, (a, b) in FIELDS.items() if f != "mag" and f in res]
        if mask_name == "frame":
            nref = ora.run_neighbours(mono, cap=True)
            pairs += [(f, res[f][off[i]:off[i + 1]].reshape(-1, 1), nref[:, c:c + 1]) for f, c in NEIGH_FIELDS.items()]
        for field, got, want in pairs:
            rtol, atol = _tol.bar(field)
            ok = ok and bool(np.all(np.isfinite(got))) and bool(np.all(np.abs(got - want) <= atol + rtol * np.abs(want)))
            e = float(_tol.rel_err(field, got, want).max())
            values += got.size
            if e > worst:
                worst, worst_field = e, field
            over = max(over, _tol.over_ceiling(field, got, want))
    return {"files": [int(t[3]) for t in targets], "frames": int(frames), "values": int(values), "max_rel_err": worst,
            "max_rel_err_descriptor": worst_field, "worst_over_ceiling": over, "passed": bool(ok), "inside_ceiling": bool(over <= 1.0),
            "against": "oracle/afx_oracle.c pipeline (LoadSample -> the per-frame descriptors of the mask: 123 spectral + 11 neighbours), bar and 10 x-observed ceiling of tests/_tol.py"}


class BatchSet:
    """The crawler's batch shape on resident data: the files cut into batches of `batch_files` (TCrawlOptions::mFilesPerBatch
    = 512), each with its own workspace and streams; one pass = every batch once, `in_flight` host threads each taking its
    share one batch at a time (enqueue, wait) the way TCrawler's workers keep one batch in flight each (Crawler.cpp of
    afec_amd/host; the reference: one task per pool thread, Crawler.cpp:706-728).  No transfers: LoadSample ran at creation."""

    def __init__(self, plan, raws, mask, batch_files, in_flight):
        from concurrent.futures import ThreadPoolExecutor
        self.first = list(range(0, len(raws), batch_files))
        self.batches = [plan.batch_from_raw(raws[i:i + batch_files], mask)[0] for i in self.first]
        self.total_frames = sum(b.total_frames for b in self.batches)
        self.in_flight = max(1, min(in_flight, len(self.batches)))
        self.pool = ThreadPoolExecutor(max_workers=self.in_flight)

    def _share(self, k):
        for b in self.batches[k::self.in_flight]:
            b.run()
            b.sync()

    def run(self):
        list(self.pool.map(self._share, range(self.in_flight)))

    def sync(self):
        pass

    def run_timed(self, steps):
        """wall milliseconds of `steps` passes (the batches run on many streams: no single pair of HIP events brackets them)"""
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run()
        return (time.perf_counter() - t0) * 1e3

    def info(self):
        return self.batches[0].info()

    def locate(self, i):
        """file i of the set -> (its batch, its index inside the batch)"""
        k = max(j for j, f in enumerate(self.first) if f <= i)
        return self.batches[k], i - self.first[k]

    def close(self):
        self.pool.shutdown()
        for b in self.batches:
            b.close()


CHAIN_MASKS = {"frame": "every per-frame low-level descriptor (the spectral set + the loop's neighbours: silence, envelope, whitening -> "
                        "spectral complexity, autocorrelation, f0, inharmonicity, tristimulus) + per-file statistics",
               "all": "the spectral low-level set of BASELINE.md C3 (MFCC, spectral statistics, flux, 28 bands, 14 sub-band descriptors, "
                      "amplitude peak / rms: AFX_D_ALL_LOW_LEVEL) + per-file statistics"}


def chain_rate(plan, workload, n_files, seed, mask_name="frame", batch_files=0, in_flight=5, device=0, clock=True, clock_hint=None):
    """BASELINE configs[2] (C3: 1 000 mono 2 s files) / configs[3] (C4: one GPU's share of 100 000 stereo 1 s files) as
    the resident chain `--workload c3|c4 --mask frame|all` times it: decoded files through LoadSample (untimed: batch
    creation), then per step the per-frame descriptors of the mask + the per-file statistics (SA:814-976, 1065).
    batch_files > 0: the crawler's shape -- batches of that many files, in_flight at a time (BatchSet) -- on the plan
    given (the crawler's pins AFX_FRAME_KERNEL_WAVE64).  Priced like the headline: frames/s x SURVEY's 5 080 algorithmic
    bytes against 8 TB/s; traffic and VALU cycles per frame from the committed profile of THIS build and configuration
    (profiles/kernel_profiles.json), null when the loaded library is another one; the VALU ceiling at the clock
    sampled during these very launches (tools/clock_probe)."""
    channels = 1 if workload == "c3" else 2
    files = make_c3_files(n_files, seed) if workload == "c3" else make_c4_files(n_files, seed)
    mask = (afx.D_ALL_PER_FRAME if mask_name == "frame" else afx.D_ALL_LOW_LEVEL) | afx.D_STATISTICS
    raws = [(f, channels) for f in files]
    if batch_files > 0:
        batch = BatchSet(plan, raws, mask, batch_files, in_flight)
    else:
        batch, _ = plan.batch_from_raw(raws, mask)
    frames = batch.total_frames
    batch.run()
    batch.sync()
    est = batch.run_timed(2) / 2
    batch.run_timed(max(3, int(40.0 / est) + 1))          # the clocks settle within ~30 ms of load (profiles/r03/clock_ramp.txt)
    steps = max(10, int(100.0 / est) + 1)
    ms = batch.run_timed(steps) / steps
    # (BatchSet: dozens of streams share the runtime's 16 hardware queues with the probe's -- its launches would wait for it)
    clock_seen = clock_during(lambda: batch.run_timed(steps), device, steps * est * 1e-3)[0] if (clock and batch_files == 0) else None
    rate = frames / (ms * 1e-3)
    info = batch.info()
    try:
        picks = sorted({0, n_files // 2, n_files - 1})
        if batch_files > 0:
            targets = [(*batch.locate(i), files[i], i) for i in picks]
        else:
            targets = [(batch, i, files[i], i) for i in picks]
        spot = chain_spot_check(targets, channels, mask_name)
    except Exception as e:  # noqa: BLE001
        spot = {"passed": False, "error": str(e)}
    n_batches, flying = (len(batch.batches), batch.in_flight) if batch_files > 0 else (1, 1)
    batch.close()
    out = {"frames_per_s": rate, "files": n_files, "frames": frames, "ms_per_step": ms, "steps": steps,
           "descriptors": CHAIN_MASKS[mask_name],
           "batches": (f"{n_batches} batches of <= {batch_files} files, {flying} in flight on their own streams (wall time per pass)"
                       if batch_files > 0 else "one batch (HIP events on its stream)"),
           "frame_kernel": {1: "wave64", 2: "halfwave"}.get(info["frame_kernel"]), "frame_kernel_class": info["feature_class"],
           "algorithmic_bytes_per_frame": SURVEY_C3_BYTES_PER_FRAME,
           "frac": rate * SURVEY_C3_BYTES_PER_FRAME / (HBM_PEAK_GBS * 1e9),
           "clock_ghz_in_run": clock_seen["clock_ghz"] if clock_seen else None,
           "traffic_ratio": None, "valu_frac": None, "kernels_ms": None, "parity_spot_check": spot}
    prof = kernel_profile("f64", mask_name, workload, "crawler" if batch_files > 0 else None)
    if prof and not prof.get("stale"):
        # (the crawler's shape: no probe beside dozens of streams, and the profile's own clock -- counters over a SUM of
        # overlapping kernels' durations -- means nothing: the clock of the same kernels as one batch, measured just before)
        ghz = clock_seen["clock_ghz"] if clock_seen else (clock_hint if clock_hint else prof["clock_ghz"])
        ceiling = 4 * 256 * ghz * 1e9 / prof["valu_cycles_per_frame"]
        out.update(traffic_ratio=prof["bytes_per_frame"] / SURVEY_C3_BYTES_PER_FRAME, valu_frac=rate / ceiling,
                   valu_frac_at_nominal_clock=rate / (4 * 256 * NOMINAL_CLOCK_GHZ * 1e9 / prof["valu_cycles_per_frame"]),
                   valu_ceiling_frames_s=ceiling, valu_clock_ghz=ghz,
                   valu_clock="sampled in this run (a repeat of the timed launches)" if clock_seen else
                   ("sampled in this run on the same files as one batch" if clock_hint else "the profiled run's"),
                   kernels_ms=prof.get("kernel_ms_per_step"), profile=prof["source"])
    return out


def secondary_rate(plan, mask, buffers, steps=5):
    """frames/s of another descriptor set on the same synthetic PCM (reported beside the headline)."""
    b = plan.batch(make_buffers(buffers, 777), mask)
    for _ in range(2):
        b.run()
    b.sync()
    ms = b.run_timed(steps) / steps
    frames = b.total_frames
    b.close()
    return frames / (ms * 1e-3)


def open_plan(device, local, world, **kw):
    """The rank's plan, or a loud end: a rank without a GPU of its own must never be counted as one (afx_plan_create
    checks the ordinal against hipGetDeviceCount and answers AFX_ERR_NO_DEVICE)."""
    try:
        return afx.Plan(device=device, **kw)
    except afx.AfxError as e:
        raise SystemExit(f"bench.py: rank with LOCAL_RANK={local} of {world} cannot use HIP device {device}: {e} "
                         f"(AFX_BENCH_DEVICE=<ordinal> pins every rank to one device: plumbing tests on a 1-GPU box)")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher.  Nothing in this process has touched the GPU (importing afec_amd loads
        # no library), the ranks are child processes, and this process only waits for them.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.fail_rank >= 0 and int(os.environ.get("RANK", "0")) == args.fail_rank:
        sys.exit(7)
    if args.stall_seconds > 0 and "AFX_BENCH_LAUNCHED_BY" in os.environ:
        time.sleep(args.stall_seconds)
    rank, world, local, dist = dist_setup(args.gpus)
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would report another job than the one asked for")

    if args.end_to_end:
        if args.workload == "c2":
            args.workload = "c4"
        device = int(os.environ.get("AFX_BENCH_DEVICE", local))
        n_files = 1000 if args.workload == "c3" else args.files
        if dist is not None:
            dist.barrier()
        st = end_to_end(args.workload, n_files, device, args.workers, 1234 + rank, repeats=max(1, args.steps // 5),
                        files_per_batch=args.files_per_batch)
        seconds, frames_all = reduce_max_sum(dist, st["seconds"], st["frames"])
        _, files_all = reduce_max_sum(dist, st["seconds"], st["files"])
        per_rank = gather_ranks(dist, {"rank": rank, "device": device, "ms_per_step": st["seconds"] * 1e3, "frames": st["frames"]})
        if rank == 0:
            print(json.dumps({
                "metric": "audio frames/sec low-level crawl, 44.1kHz 1024-hop", "value": frames_all / seconds, "unit": "frames/s",
                "n_gpus": world, "ranks_seen": [d["rank"] for d in per_rank], "ranks": per_rank,
                "steps": 1, "warmup": 0, "ms_per_step": seconds * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{args.workload.upper()} end to end: {n_files} WAV images per GPU in host memory -> streaming host "
                                       f"driver ({args.workers} workers per GPU, page-locked staging) -> LoadSample + every low-level "
                                       f"descriptor + statistics -> records back in host memory; transfers inside the timed region; "
                                       f"best of {max(1, args.steps // 5)} crawls of one process (the crawler persists between them)",
                           "files_per_s": files_all / seconds, "files_per_gpu": n_files,
                           "cold_files_per_s_per_gpu": st["files"] / st["cold_seconds"],
                           "upload_GB_per_s_per_gpu": st["pcm_bytes"] / st["seconds"] / 1e9,
                           "download_GB_per_s_per_gpu": st["result_bytes"] / st["seconds"] / 1e9,
                           "parallelism": f"replicas x{world} (files sharded i mod N, no collective)"}}))
        if dist is not None:
            dist.destroy_process_group()
        return

    # "star": the descriptors BASELINE.json's north_star names -- MFCC + spectral rms / centroid (+ spread, same
    # function) / rolloff / flatness
    star = (afx.D_MFCC | afx.D_SPECTRAL_RMS | afx.D_SPECTRAL_CENTROID | afx.D_SPECTRAL_SPREAD | afx.D_SPECTRAL_ROLLOFF |
            afx.D_SPECTRAL_FLATNESS)
    mask = {"c2": afx.D_C2, "star": star, "stats": afx.D_MFCC | afx.D_SPECTRAL_STATS,
            "all": afx.D_ALL_LOW_LEVEL, "frame": afx.D_ALL_PER_FRAME, "neighbours": afx.D_NEIGHBOURS,
            # every low-level descriptor of the reference: the per-frame ones plus the 512/128 rhythm tracker
            "everything": afx.D_ALL_PER_FRAME | afx.D_RHYTHM}[args.mask]
    precision = afx.PRECISION_F64
    # AFX_BENCH_DEVICE pins every rank to one device (plumbing tests of the N>1 path on a 1-GPU box)
    device = int(os.environ.get("AFX_BENCH_DEVICE", local))
    plan = open_plan(device, local, world, precision=precision, max_analysis_ms=0,
                    frame_kernel={"auto": afx.FRAME_KERNEL_AUTO, "wave64": afx.FRAME_KERNEL_WAVE64,
                                  "halfwave": afx.FRAME_KERNEL_HALFWAVE}[args.frame_kernel],
                    flags=afx.PLAN_NO_SIDE_STREAM if args.no_side_stream else 0)
    spot_bufs = None
    if args.workload in ("c3", "c4"):
        mask = (afx.D_ALL_PER_FRAME if args.mask in ("frame", "neighbours", "everything") else afx.D_ALL_LOW_LEVEL) | afx.D_STATISTICS
        if args.mask == "everything":
            mask |= afx.D_RHYTHM | afx.D_EFFECTIVE_LENGTH
        channels = 1 if args.workload == "c3" else 2
        bufs = make_c3_files(1000, 1234 + rank) if args.workload == "c3" else make_c4_files(args.files, 1234 + rank)
        n_bufs = len(bufs)
        if args.batch_files > 0:     # the crawler's batch shape (BatchSet): profiles of config.c4_share_at_crawler_shape
            batch = BatchSet(plan, [(b, channels) for b in bufs], mask, args.batch_files, args.in_flight)
        else:
            batch, _ = plan.batch_from_raw([(b, channels) for b in bufs], mask)
        pcm_kind = afx.PCM_F32   # the LoadSample front end keeps the float mono signal + one scale per file
    else:
        bufs = make_buffers(args.buffers, C2_SEED + DISTINCT_BUFFERS * rank)
        n_bufs = len(bufs)
        batch = plan.batch(bufs, mask)
        pcm_kind = afx.PCM_F32
        # the buffers the parity spot check looks at after the timed region: first, middle, last of the batch
        spot_bufs = {i: bufs[i] for i in sorted({0, n_bufs // 2, n_bufs - 1})}
    del bufs
    batch_info = batch.info()
    frames = batch.total_frames
    bytes_per_frame = plan.bytes_per_frame(mask & ~afx.D_STATISTICS, pcm_kind)

    tw0 = time.perf_counter()
    for _ in range(args.warmup):
        batch.run()
    batch.sync()
    step_estimate_s = (time.perf_counter() - tw0) / args.warmup if args.warmup > 0 else 1.0
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    ev_ms = batch.run_timed(args.steps)   # K launches bracketed by HIP events on the launch stream; syncs
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    # the shader clock under this load, sampled in this very process during a repeat of the K launches (clock_during says
    # why not during the timed region itself): the VALU ceiling of the roofline object is priced at the clock of this run
    clock = None
    if not args.no_clock_probe and not isinstance(batch, BatchSet):
        clock = clock_during(lambda: batch.run_timed(args.steps), device, step_estimate_s * args.steps)[0]
        if dist is not None:
            dist.barrier()
    seconds, frames_all = reduce_max_sum(dist, t1 - t0, frames)
    per_rank = gather_ranks(dist, {"rank": rank, "device": device, "ms_per_step": (t1 - t0) / args.steps * 1e3,
                                   "kernel_ms_per_step": ev_ms / args.steps, "frames": frames,
                                   "clock_ghz_in_run": clock["clock_ghz"] if clock else None})

    # What the timed launches left in HBM is checked, not only timed: the MFCC of 64 frames of three buffers of this
    # very batch against the oracle (the checker, after the timed region; tests/_tol.py's bar for the descriptor)
    spot = None
    if rank == 0 and spot_bufs and (mask & afx.D_MFCC) and not args.no_spot_check:
        try:
            spot = parity_spot_check(batch, spot_bufs, FRAMES_PER_BUFFER)
        except Exception as e:  # noqa: BLE001
            spot = {"error": str(e)}
    batch.close()
    batch = None
    headline = args.workload == "c2" and args.mask == "c2"
    secondaries = rank == 0 and world == 1 and headline and not args.no_single   # single-GPU diagnostics of the default line

    # N > 1: ONE process now drives all N devices through the C++ file-sharding crawler (the replicas above
    # scale by construction).  Every rank has given its batch back; ranks > 0 also their plan (pooled workspaces) and wait.
    sharded = None
    pinned = int(os.environ["AFX_BENCH_DEVICE"]) if "AFX_BENCH_DEVICE" in os.environ else None
    want_sharded = headline and not args.no_sharded_crawl
    if want_sharded and world > 1:
        if rank != 0:
            plan.close()
            plan = None
        dist.barrier()
        if rank == 0:
            try:
                sharded = sharded_crawl(world, args.files, 99, pinned_device=pinned)
            except Exception as e:  # noqa: BLE001
                sharded = {"error": str(e)}
        dist.barrier()

    # the literal BASELINE configs[1] shape as a secondary number: ONE resident buffer of 10 000 frames
    single = None
    if secondaries:
        one = plan.batch(make_buffers(1, C2_SEED), mask)
        for _ in range(5):
            one.run()
        one.sync()
        single = FRAMES_PER_BUFFER / (one.run_timed(50) / 50 * 1e-3)
        one.close()

    # the descriptor set north_star names and the full spectral set as secondary numbers, on the headline's batch shape
    # (--buffers x 10 000 frames: the half-wave kernels' rate depends on the batch size -- 320 000 frames: 238 M frames/s
    # for the star set, 640 000: 293 M)
    star_rate = all_rate = None
    if secondaries:
        try:   # (the headline line must not depend on the secondary batches: the full set keeps 8 KiB of magnitudes per frame)
            star_rate = secondary_rate(plan, star, args.buffers)
            all_rate = secondary_rate(plan, afx.D_ALL_LOW_LEVEL, args.buffers)
        except Exception as e:  # noqa: BLE001
            print(f"warning: secondary rates not measured: {e}", file=sys.stderr)

    # BASELINE configs[2] / configs[3] -- the crawl's own chain on decoded files -- with their own roofline and a parity
    # spot check, in the driver's line: the per-frame set incl. the loop's neighbours on one batch (round 5), and (round 6)
    # C3 on BASELINE.md's own descriptor set -- the spectral set -- and the C4 share in the shape the crawler runs it:
    # 512-file batches on the pinned 64-lane frame kernel, five in flight
    c3_chain = c4_chain = c3_spectral = c4_crawler = None
    if secondaries and not args.no_chain_rates:
        try:
            c3_chain = chain_rate(plan, "c3", 1000, 4321, device=device, clock=not args.no_clock_probe)
            c3_spectral = chain_rate(plan, "c3", 1000, 4321, mask_name="all", device=device, clock=not args.no_clock_probe)
            c4_chain = chain_rate(plan, "c4", 12500, 4321, device=device, clock=not args.no_clock_probe)
            crawler_plan = open_plan(device, local, world, precision=precision, max_analysis_ms=20000,
                                     frame_kernel=afx.FRAME_KERNEL_WAVE64)     # TSampleAnalyser's plan (afec_amd/host/Crawler.h)
            try:
                c4_crawler = chain_rate(crawler_plan, "c4", 12500, 4321, batch_files=512, in_flight=5, device=device, clock=not args.no_clock_probe,
                                        clock_hint=c4_chain.get("clock_ghz_in_run") if c4_chain else None)
            finally:
                crawler_plan.close()
        except Exception as e:  # noqa: BLE001
            print(f"warning: chain rates not measured: {e}", file=sys.stderr)

    # the streaming host driver on C4's per-GPU share, every transfer inside the timed region (secondary number)
    e2e = None
    if secondaries:
        try:
            import tempfile
            from afec_amd import hostlib
            W = 5     # host threads per GPU of every crawl below unless its object says otherwise (TCrawlOptions' choice for one GPU)
            st = end_to_end("c4", 12500, device, W, 99, repeats=8, files_per_batch=512)
            e2e = {"workload": "C4 share: 12 500 stereo 1.0 s 16-bit WAV images in host memory -> RIFF parse -> page-locked staging -> "
                               "upload -> LoadSample + every low-level descriptor (per-frame set and rhythm tracker) + statistics -> records back in host memory",
                   "files_per_s": st["files"] / st["seconds"], "frames_per_s": st["frames"] / st["seconds"],
                   "cold_files_per_s": st["files"] / st["cold_seconds"],   # first crawl of the process: page-locked and device pools empty
                   "upload_GB_per_s": st["pcm_bytes"] / st["seconds"] / 1e9, "download_GB_per_s": st["result_bytes"] / st["seconds"] / 1e9,
                   # 55 GB/s: one page-locked upload stream on this host link (tools/link_rate.py, profiles/r02/README.md)
                   "upload_frac_of_host_link": st["pcm_bytes"] / st["seconds"] / 55e9,
                   "busy_host_cpus": st["busy_cpus_median"],     # median over the warm crawls of this process
                   "busy_host_cpus_of_the_fastest_crawl": st["cpu_seconds"] / st["seconds"],
                   "workers": W, "files_per_batch": 512}
            # the same crawl with the single writer inserting every file into the reference's sqlite `assets` table
            # (461 columns, ~68 KB of msgpack per one-second file; one transaction per batch of 512 files), database on
            # tmpfs: the writer, not the GPU, bounds it
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            with tempfile.TemporaryDirectory(dir=shm) as td:
                sd = end_to_end("c4", 4096, device, W, 99, database=os.path.join(td, "afec-ll.db"), repeats=3, files_per_batch=512)
                e2e["with_database"] = {
                    "files_per_s": sd["files"] / sd["seconds"],
                    "writer_files_per_s": sd["files"] / sd["writer_seconds"] if sd["writer_seconds"] else None, "workers": W,
                    "database": "sqlite `assets` table on tmpfs, 4096 files per crawl, best of 3 crawls, each into a new file (the first one also allocates the workers' row buffers)"}
                # the same with TCrawlOptions::mDatabasePragmas: 64 KiB pages, journal in memory, no fsync per commit
                try:
                    hostlib.set_database_pragmas("PRAGMA page_size=65536; PRAGMA journal_mode=MEMORY; PRAGMA synchronous=OFF")
                    sp = end_to_end("c4", 4096, device, W, 99, database=os.path.join(td, "afec-ll-tuned.db"), repeats=2, files_per_batch=512)
                    e2e["with_database"]["files_per_s_with_pragmas"] = sp["files"] / sp["seconds"]
                finally:
                    hostlib.set_database_pragmas("")
            # the crawler reading the files itself (what a crawl of a sample library does): 12 500 files on tmpfs, the data
            # chunks pread straight into the page-locked staging buffers; eight workers: the kernel's copy out of the page
            # cache is the cost there, and more readers help (Crawler.h)
            with tempfile.TemporaryDirectory(dir=shm) as td:
                sf = end_to_end("c4", 12500, device, 8, 99, repeats=3, files_per_batch=512, files_dir=td)
                e2e["from_files_on_tmpfs"] = {"files_per_s": sf["files"] / sf["seconds"], "busy_host_cpus": sf["busy_cpus_median"], "workers": 8}
            # the same 12 500 files labelled 48 kHz: every one goes through the sample-rate conversion on the GPU first
            # (libresample's arithmetic, afx_resample.hip), then the same pipeline on 0.92 x the samples
            sr = end_to_end("c4", 12500, device, W, 99, repeats=3, files_per_batch=512, rate=48000)
            e2e["at_48_kHz"] = {"files_per_s": sr["files"] / sr["seconds"], "workers": W}
            # last (its crawler -- eight analysers with their plans, pools and page-locked staging on this one device -- stays
            # resident once built): the same share as eight shards with one worker each, what one process driving the 8
            # GPUs of a node with a worker per GPU costs the host (Crawler.cpp:706-728; the shards share the device here)
            try:
                s8 = None
                for _ in range(3):
                    t8 = hostlib.crawl(_c4_images(12500, 99), devices=(device,) * 8, workers=1, files_per_batch=512)
                    if s8 is None or t8["seconds"] < s8["seconds"]:
                        s8 = t8
                e2e["eight_shards_one_worker"] = {"files_per_s": s8["files"] / s8["seconds"], "busy_host_cpus": s8["cpu_seconds"] / s8["seconds"],
                                                  "workers": 1, "shards": 8}
            except Exception as e8:  # noqa: BLE001
                e2e["eight_shards_one_worker"] = {"error": str(e8)}
        except Exception as e:  # noqa: BLE001  (the headline must not depend on the host library)
            e2e = dict(e2e or {}, error=str(e))

    # N = 1: the sharded crawl is the same driver over one device; it runs last, from a cold crawler (the crawlers the
    # measurements above left resident are dropped first), and must agree with end_to_end_host_driver.files_per_s
    if want_sharded and world == 1 and rank == 0 and not args.no_single:
        try:
            from afec_amd import hostlib
            hostlib.release()
            sharded = sharded_crawl(1, args.files, 99, pinned_device=pinned if pinned is not None else device, repeats=8)   # (as many crawls as end_to_end_host_driver took its best of)
        except Exception as e:  # noqa: BLE001
            sharded = {"error": str(e)}

    if rank == 0:
        launch_ms = ev_ms / args.steps
        achieved = bytes_per_frame * frames / (launch_ms * 1e-3) / 1e9
        prof = kernel_profile(args.precision, args.mask, args.workload, "crawler" if args.batch_files > 0 else None)
        stale_profile = prof if (prof and prof.get("stale")) else None
        if stale_profile:
            prof = None
        valu = None
        if prof and prof.get("valu_cycles_per_frame"):
            # f64 VALU ceiling of this instruction mix: 4 SIMDs x 256 CUs x clock / pipe cycles per frame.  The cycles per
            # frame are a property of the build (counters of the committed profile, quoted only for the loaded build's
            # hash); the clock is the one sampled DURING the timed launches of this run (the profiled run's when the probe
            # is off), and the same rate is also priced at the part's nominal 2.4 GHz
            rate = frames / (launch_ms * 1e-3)
            ghz = clock["clock_ghz"] if clock else prof["clock_ghz"]
            ceiling = 4 * 256 * ghz * 1e9 / prof["valu_cycles_per_frame"]
            nominal = 4 * 256 * NOMINAL_CLOCK_GHZ * 1e9 / prof["valu_cycles_per_frame"]
            valu = {"cycles_per_frame": prof["valu_cycles_per_frame"], "instructions_per_frame": prof["valu_instructions_per_frame"],
                    "clock_ghz": ghz, "clock_source": "sampled in this run, during a repeat of the timed launches (tools/clock_probe)" if clock
                    else "GRBM_GUI_ACTIVE of the profiled run (no probe in this run)",
                    "clock_ghz_profiled_run": prof["clock_ghz"], "ceiling_frames_s": ceiling, "frac": rate / ceiling,
                    "nominal_clock_ghz": NOMINAL_CLOCK_GHZ, "ceiling_frames_s_at_nominal_clock": nominal,
                    "frac_at_nominal_clock": rate / nominal, "source": prof["source"]}
        out = {
            "metric": "audio frames/sec low-level crawl, 44.1kHz 1024-hop",
            "value": frames_all * args.steps / seconds,
            "unit": "frames/s",
            "n_gpus": world,
            "ranks_seen": [d["rank"] for d in per_rank],
            "ranks": per_rank,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": seconds / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": (f"C3: {'every' if args.mask == 'everything' else 'every per-frame' if args.mask in ('frame', 'neighbours') else 'spectral'} low-level "
                             f"descriptor set + per-file statistics, {n_bufs} synthetic 2.0 s mono 16-bit files per "
                             f"GPU through the LoadSample front end") if args.workload == "c3" else
                            (f"C4 share: {'every' if args.mask == 'everything' else 'every per-frame' if args.mask in ('frame', 'neighbours') else 'spectral'} low-level "
                             f"descriptor set + per-file statistics, {n_bufs} synthetic 1.0 s stereo 16-bit files per GPU "
                             f"through the LoadSample front end") if args.workload == "c4" else
                            (f"C2 x{args.buffers}: 2048/1024 STFT + 14-coef MFCC, {args.buffers} mono float32 "
                             f"buffers of {FRAMES_PER_BUFFER} frames per GPU, U(-1,1) from std::mt19937 + std::uniform_real_distribution<float> "
                             f"(SURVEY 8d's generator; buffer b of rank r seeded 1234 + 64 r + b: buffer 0 of rank 0 is mt19937(1234) itself)"
                             if args.mask == "c2" else f"{args.mask} descriptor set, {args.buffers} x {FRAMES_PER_BUFFER} frames"),
                "frames_per_gpu_per_step": frames,
                "files_per_gpu_per_step": n_bufs,
                "parity_spot_check": spot,
                "frame_kernel": {1: "wave64", 2: "halfwave"}.get(batch_info["frame_kernel"]),
                "frame_kernel_class": batch_info["feature_class"],
                "chunk_frames": batch_info["chunk_frames"],
                "single_10k_frame_buffer_frames_per_s": single,
                "star_descriptor_set_frames_per_s": star_rate,
                "all_spectral_descriptors_frames_per_s": all_rate,
                "c3_frames_per_s": c3_chain,
                "c3_spectral_set_frames_per_s": c3_spectral,
                "c4_share_frames_per_s": c4_chain,
                "c4_share_at_crawler_shape": c4_crawler,
                "end_to_end_host_driver": e2e,
                "sharded_crawl": sharded,
                "pcm": "f32 resident in HBM" if args.workload == "c2" else
                       "f32 mono signal + one double scale per file (LoadSample output, SampleAnalyser.cpp:710-718) resident in HBM",
                "parallelism": f"replicas x{world} (buffers sharded, no collective); config.sharded_crawl: one process, "
                               f"files i mod {world} over {world} device(s)",
            },
            # achieved / peak / frac / traffic: the HBM roofline the path is priced against (a streaming scan of PCM,
            # SURVEY 8d), also as the "hbm" object.  "bound": the resource the counters say binds these f64 kernels --
            # the vector ALU ("valu_f64", with its own ceiling in "valu") -- or "hbm" when no current profile says so.
            "roofline": {
                "bound": "valu_f64" if valu else "hbm",
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None if not prof else prof["bytes_per_frame"] * frames},
                "profile_build": afx.build_info() if prof else None,
                "profile_stale": (stale_profile["measured_on"] or "a build that recorded no hash") if stale_profile else None,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "peak_achievable_read": HBM_ACHIEVABLE_READ_GBS,
                "frac_of_achievable": achieved / HBM_ACHIEVABLE_READ_GBS,
                "traffic": None if not prof else prof["bytes_per_frame"] * frames,
                "kernel": prof["dominant_kernel"] if prof else None,
                "kernels_timed": prof["kernels"] if prof else None,
                "limiter": prof["limiter"] if prof else None,
                "valu": valu,
                "clock_ghz_in_run": clock["clock_ghz"] if clock else None,
                "clock_probe": clock,
                "algorithmic_bytes_per_frame": bytes_per_frame,
                "launch_ms": launch_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            t_cpu = time.time()
            out["cpu_baseline"] = cpu_baseline(args.cpu_frames)
            out["cpu_baseline"]["seconds"] = time.time() - t_cpu
        # wall time of this whole run as rank 0 saw it (imports, input generation, the timed region, every secondary
        # object of the line, the CPU baseline): what the driver's own clock around the command should read, minus the
        # interpreter's start-up -- the default run has to finish within minutes
        out["run_seconds"] = time.time() - T_SCRIPT_START
        print(json.dumps(out))
    if plan is not None:
        plan.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
