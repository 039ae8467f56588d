"""Edge cases and concurrency of the C-ABI on the GPU."""
import threading
import time

import numpy as np
import pytest

import afec_amd as afx
from tests import _tol
from tests._oracle import Oracle

pytestmark = pytest.mark.gpu


def test_empty_batch_and_all_short_buffers():
    plan = afx.Plan()
    res = plan.extract([], afx.D_MFCC)
    assert res["mfcc"].shape == (0, 14) and res["frame_offset"].tolist() == [0]
    res = plan.extract([np.zeros(10, np.float32), np.zeros(2047, np.float32)], afx.D_ALL_LOW_LEVEL)
    assert res["mfcc"].shape == (0, 14) and res["frame_offset"].tolist() == [0, 0, 0]
    b = plan.batch([np.zeros(5, np.float32)], afx.D_MFCC | afx.D_STATISTICS)
    b.run()
    st = b.fetch_statistics()
    assert np.all(st["mfcc"] == 0.0)          # Length == 0: TStatistics::Calc assigns zeros
    b.close()
    plan.close()


def test_concurrent_calls_on_one_plan_match_serial():
    """The plan is shared by worker threads like the const TSampleAnalyser (Crawler.cpp:599, 706-728)."""
    rng = np.random.default_rng(41)
    inputs = [[rng.uniform(-1, 1, 2048 + 1024 * int(rng.integers(1, 60))).astype(np.float32) for _ in range(5)]
              for _ in range(8)]
    plan = afx.Plan()
    mask = afx.D_ALL_PER_FRAME | afx.D_EFFECTIVE_LENGTH       # every kernel of the library
    serial = [plan.extract(bufs, mask) for bufs in inputs]
    raws = [[(np.round(b * 20000).astype(np.int16), 1) for b in bufs] for bufs in inputs]
    out = [None] * len(inputs)
    errs = []

    def work(i):
        try:
            for _ in range(3):
                out[i] = plan.extract(inputs[i], mask)
                # and the LoadSample front end with its per-call staging, from the same threads
                b, _ = plan.batch_from_raw(raws[i], afx.D_MFCC | afx.D_STATISTICS)
                b.run()
                b.fetch_statistics()
                b.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for a, b in zip(serial, out):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k])
    plan.close()


def test_long_uncapped_buffer_head_and_tail():
    rng = np.random.default_rng(42)
    nf = 70000                                   # > 2 x 32-frame chunks per wave, ~287 MB of f32 PCM
    x = rng.uniform(-1, 1, 2048 + 1024 * (nf - 1)).astype(np.float32)
    plan = afx.Plan(max_analysis_ms=0)
    res = plan.extract([x], afx.D_MFCC | afx.D_SPECTRAL_ROLLOFF | afx.D_SPECTRAL_FLUX)
    assert res["mfcc"].shape == (nf, 14)
    ora = Oracle()
    for f0 in (0, 31, 32, 33, nf - 40):
        seg = x[f0 * 1024: f0 * 1024 + 2048 + 1024 * 7].astype(np.float64)
        want = ora.run(seg)
        _tol.check_gpu("mfcc", res["mfcc"][f0:f0 + 8], want[:, 1024:1038], *_tol.GPU_TOL["mfcc"], what=f"f0={f0} ")
        np.testing.assert_array_equal(res["spectral_rolloff"][f0:f0 + 8], want[:, 1043])
        # flux of a frame depends on the previous frame of the same buffer, not of the slice
        if f0 == 0:
            _tol.check_gpu("spectral_flux", res["spectral_flux"][:8], want[:, 1045], *_tol.GPU_TOL["spectral_flux"])
        else:
            _tol.check_gpu("spectral_flux", res["spectral_flux"][f0 + 1:f0 + 8], want[1:, 1045], *_tol.GPU_TOL["spectral_flux"])
    plan.close()


def test_pcie_inclusive_one_shot_rate_is_reported():
    """HISTORY.md section 7 quotes the PCIe-inclusive afx_extract_batch rate; keep it measurable."""
    rng = np.random.default_rng(43)
    x = rng.uniform(-1, 1, 2048 + 1024 * 9999).astype(np.float32)
    plan = afx.Plan(max_analysis_ms=0)
    plan.extract([x], afx.D_C2)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        plan.extract([x], afx.D_C2)
    dt = (time.perf_counter() - t0) / n
    rate = 10000 / dt
    print(f"one-shot afx_extract_batch, 10k frames incl. H2D/D2H and allocation: {rate / 1e6:.2f} M frames/s")
    assert rate > 1e5
    plan.close()


@pytest.mark.parametrize("kernel", ["wave64", "halfwave"])
def test_large_call_is_split_internally_and_matches_small_calls(kernel):
    """afx_extract_batch cuts calls of more than 2^19 frames into groups of buffers: the same results, bit for bit, as
    small calls on the same STFT kernel (the layout is pinned: by batch size a default plan would give the groups the
    half-wave kernel and the small call the 64-lane one, which agree to rounding only)."""
    rng = np.random.default_rng(44)
    base = [rng.uniform(-1, 1, 2048 + 1024 * int(n)).astype(np.float32) for n in (700, 1, 859, 300, 0, 512)]
    bufs = [base[i % len(base)] for i in range(1400)] + [np.zeros(10, np.float32)]   # ~550k frames
    plan = afx.Plan(frame_kernel=afx.FRAME_KERNEL_WAVE64 if kernel == "wave64" else afx.FRAME_KERNEL_HALFWAVE)
    mask = afx.D_MFCC | afx.D_SPECTRAL_FLUX | afx.D_SPECTRAL_ROLLOFF
    big = plan.extract(bufs, mask)
    ref = plan.extract(base, mask)
    assert big["frame_offset"][-1] > (1 << 19)
    assert big["buf_status"].tolist() == [0] * len(bufs)
    roff = ref["frame_offset"]
    for i in (0, 1, 5, 6, 700, 1393, 1399):
        j = i % len(base)
        a, b = big["frame_offset"][i], big["frame_offset"][i + 1]
        assert b - a == roff[j + 1] - roff[j]
        for k in ("mfcc", "spectral_flux", "spectral_rolloff"):
            np.testing.assert_array_equal(big[k][a:b], ref[k][roff[j]:roff[j + 1]])
    plan.close()


def test_vanishing_amplitudes_flush_like_the_reference():
    """a decay down to 1e-160: TAudioMath::Magnitude runs with DAZ + FZ set, so bins below ~1.5e-154 are exactly
    zero there (flatness 1.0, no peaks); the kernel flushes the same way and stays finite all the way down"""
    rng = np.random.default_rng(51)
    n = 2048 + 1024 * 50
    x = rng.standard_normal(n) * np.exp(-np.arange(n) / 120.0)        # 1e-150 after ~41 000 samples, 1e-190 at the end
    plan = afx.Plan(max_analysis_ms=0)
    res = plan.extract([x], afx.D_ALL_LOW_LEVEL)
    ref = Oracle().run(x)
    from tests._oracle import FIELDS
    for field, (a, b) in FIELDS.items():
        if field == "mag":
            continue
        got = res[field].reshape(ref.shape[0], -1)
        assert np.all(np.isfinite(got)), field
        rtol, atol = _tol.GPU_TOL[field]
        # the transition (per-product flush there, flush of the sum here: |X| around 1e-154) is left out
        keep = np.r_[0:37, 46:51]
        _tol.check_gpu(field, got[keep], ref[keep, a:b], rtol, atol, what="vanishing ")
    assert res["spectral_flatness"][-1] == 1.0 and not res["sub_complexity"][-1].any()
    plan.close()


def test_fetch_before_run_is_refused_and_outputs_are_validated():
    """Workspaces are pooled: a fetch before afx_batch_run would hand back another batch's results."""
    plan = afx.Plan()
    x = np.random.default_rng(3).uniform(-1, 1, 2048 + 1024 * 5).astype(np.float32)
    plan.extract([x], afx.D_ALL_LOW_LEVEL)            # leaves results in the pooled workspace
    b = plan.batch([x], afx.D_MFCC | afx.D_STATISTICS)
    with pytest.raises(afx.AfxError) as ei:
        b.fetch()
    assert ei.value.status == -1
    with pytest.raises(afx.AfxError):
        b.fetch_statistics()
    b.run()
    assert b.fetch()["mfcc"].shape == (6, 14)
    b.close()
    plan.close()


def test_plan_destroyed_before_its_batch():
    """afx_plan_destroy is deferred while batches of the plan are alive (interpreter exit order, user code)."""
    plan = afx.Plan()
    x = np.random.default_rng(4).uniform(-1, 1, 2048 + 1024 * 9).astype(np.float32)
    want = plan.extract([x], afx.D_MFCC)["mfcc"].copy()
    b = plan.batch([x], afx.D_MFCC)
    L, handle = plan.L, plan.h
    plan.h = None                      # keep Plan.close() from touching it again
    L.afx_plan_destroy(handle)         # the C-ABI call, with the batch still alive
    b.run()
    np.testing.assert_array_equal(b.fetch()["mfcc"], want)
    b.close()                          # the last reference frees the plan


def test_mixed_pcm_types_in_one_call_flag_the_minority():
    plan = afx.Plan()
    rng = np.random.default_rng(6)
    a = rng.uniform(-1, 1, 2048 + 1024 * 3).astype(np.float32)
    d = rng.uniform(-1, 1, 2048 + 1024 * 2)
    res = plan.extract([a, d, a], afx.D_MFCC)
    assert res["buf_status"].tolist() == [0, -6, 0]
    assert res["frame_offset"].tolist() == [0, 4, 4, 8]
    plan.close()


def test_rhythm_calls_are_validated_and_threads_agree():
    """Error behaviour of the rhythm entry points, and the rhythm tracker from several threads on one plan (the side
    stream, the plan's shared transfer streams and the pooled workspaces are shared state)."""
    from tests import _oracle
    rng = np.random.default_rng(77)
    plan = afx.Plan()
    x = (rng.uniform(-1, 1, 60000) * np.exp(-np.arange(60000) / 20000.0)).astype(np.float32)
    b = plan.batch([x], afx.D_MFCC)
    b.run()
    with pytest.raises(afx.AfxError):          # AFX_D_RHYTHM was not in the mask
        b.fetch_rhythm()
    with pytest.raises(afx.AfxError):
        b.set_file_info([(44100, 0, x.size)])
    assert b.rhythm_frames().tolist() == [0, 0]
    b.close()
    b = plan.batch([x], afx.D_RHYTHM)
    with pytest.raises(afx.AfxError):          # fetch before run
        b.fetch_rhythm()
    b.run()
    with pytest.raises(afx.AfxError):          # onset statistics need AFX_D_STATISTICS
        b.fetch_rhythm(statistics=True)
    want = b.fetch_rhythm(onset_functions=True)
    b.close()

    files = [[(rng.uniform(-1, 1, 30000 + 4000 * k) * np.exp(-np.arange(30000 + 4000 * k) / 9000.0)).astype(np.float32)
              for k in range(6)] for _ in range(6)]
    serial = []
    for bufs in files:
        bb = plan.batch(bufs, afx.D_RHYTHM | afx.D_ALL_PER_FRAME | afx.D_STATISTICS)
        bb.run()
        serial.append((bb.fetch_rhythm(statistics=True), bb.fetch()))
        bb.close()
    out, errs = [None] * len(files), []

    def work(i):
        try:
            for _ in range(3):
                bb = plan.batch(files[i], afx.D_RHYTHM | afx.D_ALL_PER_FRAME | afx.D_STATISTICS)
                bb.run()
                out[i] = (bb.fetch_rhythm(statistics=True), bb.fetch())
                bb.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(files))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for (r0, f0), (r1, f1) in zip(serial, out):
        for k in ("onsets", "scalars", "onset_statistics"):
            assert np.array_equal(r0[k], r1[k]), k
        assert np.array_equal(f0["mfcc"], f1["mfcc"]) and np.array_equal(f0["f0"], f1["f0"])
    # the single-file result did not depend on what ran beside it either
    ref = _oracle.Oracle().run_rhythm(x.astype(np.float64), cap=True)
    assert np.array_equal(np.nonzero(want["onsets"][:, 0])[0], np.nonzero(ref["onsets"][0])[0])
    plan.close()


def test_page_locked_blocks_of_every_size_class():
    """afx_host_alloc: blocks of 8 MiB and more are huge-page mappings registered with the runtime, smaller ones come
    from hipHostMalloc; both are plain memory to the host and sources of direct transfers to the device."""
    from afec_amd import capi
    rng = np.random.default_rng(5)
    owners = []
    for n_bytes in (1, 4096, (8 << 20) - 1, 8 << 20, (8 << 20) + 12345, 24 << 20):
        arr, owner = capi.pinned_array((n_bytes,), np.uint8)
        arr[:] = 0x5A
        arr[-1] = 0xA5
        assert int(arr[-1]) == 0xA5 and arr.size == n_bytes and (n_bytes == 1 or int(arr[0]) == 0x5A)
        owners.append((arr, owner))
    del owners                                   # freed in any order
    x = (0.5 * rng.standard_normal(2048 + 1024 * 4999)).astype(np.float32)      # 20 MiB of PCM
    pinned, owner = capi.pinned_array(x.shape, np.float32)
    pinned[:] = x
    p = afx.Plan()
    a = p.extract([x], afx.D_MFCC | afx.D_SPECTRAL_CENTROID)
    b = p.extract([pinned], afx.D_MFCC | afx.D_SPECTRAL_CENTROID)
    np.testing.assert_array_equal(a["mfcc"], b["mfcc"])
    np.testing.assert_array_equal(a["spectral_centroid"], b["spectral_centroid"])
    p.close()
    del pinned, owner


def test_side_stream_and_queue_order_do_not_change_a_result():
    """The time-domain kernels and the rhythm tracker run beside the spectral chain on the workspace's side stream, and the
    time-domain / band / whitening kernels draw their chunks from device work queues whose order differs from run to run:
    the same batch on a plan with AFX_PLAN_NO_SIDE_STREAM, and run again and again on one plan, gives the same bits."""
    rng = np.random.default_rng(91)
    t = np.arange(44100)
    pool = [np.round(9000 * rng.uniform(-1, 1, 44100)).astype(np.int16),
            np.round(12000 * np.sin(2 * np.pi * 330 * t / 44100) * np.exp(-t / 15000.0)).astype(np.int16),
            np.round(4000 * rng.standard_normal(44100) * (t % 8000 < 1500)).astype(np.int16)]
    raws = [(pool[i % 3], 1) for i in range(700)]
    results = []
    for flags in (0, afx.PLAN_NO_SIDE_STREAM):
        plan = afx.Plan(flags=flags)
        for mask in (afx.D_ALL_PER_FRAME | afx.D_STATISTICS, afx.D_ALL_PER_FRAME | afx.D_STATISTICS | afx.D_RHYTHM):
            b, _ = plan.batch_from_raw(raws, mask)
            runs = []
            for _ in range(3):
                b.run()
                res, st = b.fetch(), b.fetch_statistics()
                res.update({"stat_" + k: v for k, v in st.items()})
                if mask & afx.D_RHYTHM:
                    r = b.fetch_rhythm(onset_functions=True)
                    res.update({"rhythm_" + k: v for k, v in r.items() if isinstance(v, np.ndarray)})
                runs.append(res)
            for k in runs[0]:
                np.testing.assert_array_equal(runs[0][k], runs[1][k], err_msg=k)
                np.testing.assert_array_equal(runs[0][k], runs[2][k], err_msg=k)
            results.append(runs[0])
            b.close()
        plan.close()
    for with_side, without in ((results[0], results[2]), (results[1], results[3])):
        assert with_side.keys() == without.keys()
        for k in with_side:
            np.testing.assert_array_equal(with_side[k], without[k], err_msg=k)


def test_an_allocation_that_finds_the_device_full_gets_the_idle_pool_back_and_the_probe_says_alive():
    """Idle pooled workspaces keep their capacity (up to 16 per plan): a batch that failed for memory would fail again,
    and the device probe -- which used to hipMalloc(256) -- could declare a live device lost.  Three batches leave ~10 GB
    idle in the pool, a dummy allocation takes all other free device memory, and a batch that needs ~7 GB is created: the
    first hipMalloc fails, the library gives the idle workspaces back (afx_workspace.cpp: ws_reserve -> pool_trim) and the
    batch comes to be and computes the right thing; afx_plan_probe_device answers AFX_OK with the device that full.
    (Workspaces above 4 GiB are not pooled: the three held here are 3.1 GB each.)"""
    import ctypes
    import os
    plan = afx.Plan(max_analysis_ms=0)
    # the HIP runtime libafx_hip.so itself runs on (a process that imported torch has torch's own copy loaded as well, and
    # that one has no device): the one of the image's ROCm, by path
    rocm = "/opt/rocm/lib/libamdhip64.so"
    hip = ctypes.CDLL(rocm if os.path.exists(rocm) else "libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) != 0:
        pytest.skip("no handle on the HIP runtime the library uses")
    L = plan.L
    rng = np.random.default_rng(17)
    frames = 200_000
    x = rng.uniform(-1, 1, (frames - 1) * 1024 + 2048).astype(np.float32)      # 0.8 GB of PCM, 1.6 GB of magnitudes
    mask = afx.D_MFCC | afx.D_SPECTRAL_FLUX        # flux is taken from stored magnitudes: 8 KiB per frame in the workspace
    held = [plan.batch([x], mask) for _ in range(3)]                            # three workspaces at once ...
    want = None
    for b in held:
        b.run()
    want = held[0].fetch()["mfcc"][:64].copy()
    for b in held:
        b.close()                                                               # ... idle in the plan's pool now
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    idle_free = free.value
    dummy = ctypes.c_void_p()
    leave = 5 << 29                                                             # 2.5 GB: less than the ~5 GB the next batch needs
    assert hip.hipMalloc(ctypes.byref(dummy), ctypes.c_size_t(idle_free - leave)) == 0
    try:
        assert L.afx_plan_probe_device(plan.h) == 0                            # no allocation in the probe
        hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        before = free.value
        assert before < (3 << 30)
        big = plan.batch([x, x], mask)                                          # 2 x: does not fit what is free, fits after the trim
        hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        assert free.value > before + (1 << 30)                                  # more is free with the batch alive than before it: the idle pool went back
        big.run()
        got = big.fetch()["mfcc"]
        np.testing.assert_array_equal(got[:64], want)
        np.testing.assert_array_equal(got[frames:frames + 64], want)
        assert L.afx_plan_probe_device(plan.h) == 0
        big.close()
    finally:
        hip.hipFree(dummy)
    plan.close()


def test_a_refused_first_file_does_not_move_the_upload_of_the_one_behind_it():
    """Round 6, found on the CPU by the sanitizer harness (tests/sanitize/fuzz_host_abi.cpp): a file the conversion planner
    refuses (a header above 16 x the analyser's rate) keeps its place in the raw arena's layout; with exactly one valid
    file behind it -- its own scattered buffer -- the one-transfer upload used to start `raw_off` bytes in FRONT of that
    buffer (a read outside the caller's memory, and the valid file's samples landed shifted).  The valid file's results
    must be those of the same file analysed alone."""
    rng = np.random.default_rng(77)
    bad = (rng.integers(-20000, 20000, 50000)).astype(np.int16)
    good = np.round(15000.0 * np.sin(2 * np.pi * 523.0 * np.arange(30000) / 44100.0) * np.exp(-np.arange(30000) / 9000.0)).astype(np.int16)
    plan = afx.Plan()
    mask = afx.D_ALL_LOW_LEVEL
    alone, _ = plan.batch_from_raw([(good, 1)], mask)
    alone.run()
    want = alone.fetch()
    both, info = plan.batch_from_raw([(bad, 1, 800000), (good, 1)], mask)
    both.run()
    got = both.fetch()
    assert got["buf_status"][0] != 0 and got["buf_status"][1] == 0
    assert got["frame_offset"].tolist() == [0, 0, want["frame_offset"][1]]
    for k in ("mfcc", "spectral_centroid", "spectrum_bands", "sub_contrast", "amplitude_peak"):
        np.testing.assert_array_equal(got[k], want[k])
    alone.close(); both.close(); plan.close()
