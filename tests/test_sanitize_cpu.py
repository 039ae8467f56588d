"""The mock device of tests/sanitize (a host-memory stand-in for the HIP runtime + mock kernels that assert the chunk-table,
queue and placement invariants) as plain CPU tests: the C-ABI's host code and the sharded crawler run on it without a GPU.
The sanitizer builds of the same programs are tools/sanitize_cpu.sh (profiles/r06/sanitize_cpu.txt, tsan_crawler.txt)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plain_builds():
    out = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "build.sh"), "plain"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    return {os.path.basename(p).replace("_plain", ""): p for p in out.stdout.decode().split()}


def test_files_are_sharded_i_mod_g_and_rows_do_not_depend_on_the_device(plain_builds):
    """The assertion of tests/test_gpu_crawler.py's two-GPU test (skipped on every 1-GPU box) on 1, 2 and 8 MOCK devices:
    afec::TCrawler analyses file i on device i mod G (Crawler.cpp:706-728 one level up), delivers every file once, the
    same content has the same row digest on every device and in every batch; injected failures, a lost device and an
    external abort behave as the reference's per-file try / catch and SIGINT flag do (SampleAnalyser.cpp:368-408,
    Crawler.cpp:69-73, 717-720).  The program aborts on the first violated check."""
    r = subprocess.run([plain_builds["tsan_crawler"], "420"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    text = r.stdout.decode()
    assert r.returncode == 0, r.stderr.decode()[-3000:] + text[-2000:]
    for g in (1, 2, 8):
        assert f"G = {g}: 420 files" in text
    assert "external abort" in text and text.strip().endswith("tsan_crawler: clean")


def test_the_c_abi_host_code_on_the_mock_device(plain_builds):
    """fuzzed ragged batches through include/afx.h: 0-frame buffers, a 2^30-sample claim behind the 20 s cap, >= 768-buffer
    batches (whole-file whitening chunks), refused conversions, allocation failures; every invariant of mock_kernels.cpp"""
    r = subprocess.run([plain_builds["fuzz_host_abi"], "120", "11", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"clean" in r.stdout


def test_the_mock_restates_the_launch_rules_of_the_product_library():
    """frames_use_halfwave / frames32_class / frames_feature_class / ... live in the .hip files; mock_kernels.cpp restates
    them.  The product library can be loaded without a GPU: both must agree on every mask the planner can produce."""
    import afec_amd
    real = ctypes.CDLL(afec_amd.library_path())
    so = "/tmp/afx_san/libmock_rules.so"
    os.makedirs("/tmp/afx_san", exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "sanitize", "hipstub"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "sanitize", "mock_kernels.cpp"),
                           os.path.join(ROOT, "tests", "sanitize", "hipstub", "hip_stub.cpp"), "-o", so])
    mock = ctypes.CDLL(so)
    names = {"_ZN3afx19frames_use_halfwaveEjii": (ctypes.c_bool, [ctypes.c_uint32, ctypes.c_int, ctypes.c_int]),
             "_ZN3afx14frames32_classEj": (ctypes.c_int, [ctypes.c_uint32]),
             "_ZN3afx20frames_feature_classEj": (ctypes.c_int, [ctypes.c_uint32]),
             "_ZN3afx22frames_waves_per_blockEj": (ctypes.c_int, [ctypes.c_uint32]),
             "_ZN3afx24frames32_waves_per_blockEv": (ctypes.c_int, []),
             "_ZN3afx25frames32_stat_tmp_doublesEv": (ctypes.c_int, []),
             "_ZN3afx25load_scan_blocks_per_fileEi": (ctypes.c_int, [ctypes.c_int]),
             "_ZN3afx15resample_blocksEl": (ctypes.c_int64, [ctypes.c_int64])}
    for lib in (real, mock):
        for n, (res, args) in names.items():
            getattr(lib, n).restype = res
            getattr(lib, n).argtypes = args
    extra = [0, 1 << 30, 1 << 31, (1 << 30) | (1 << 31)]
    masks = sorted({(m & 0x3FFF) | e for m in list(range(0, 0x4000, 7)) + [1, 0xFF, 0x1FFF, 0x3FFF, 0x2001] for e in extra})
    for m in masks:
        assert real._ZN3afx14frames32_classEj(m) == mock._ZN3afx14frames32_classEj(m), hex(m)
        assert real._ZN3afx20frames_feature_classEj(m) == mock._ZN3afx20frames_feature_classEj(m), hex(m)
        assert real._ZN3afx22frames_waves_per_blockEj(m) == mock._ZN3afx22frames_waves_per_blockEj(m), hex(m)
        for dtype in (0, 1, 2):
            assert bool(real._ZN3afx19frames_use_halfwaveEjii(m, 0, dtype)) == bool(mock._ZN3afx19frames_use_halfwaveEjii(m, 0, dtype)), (hex(m), dtype)
    assert real._ZN3afx24frames32_waves_per_blockEv() == mock._ZN3afx24frames32_waves_per_blockEv()
    assert real._ZN3afx25frames32_stat_tmp_doublesEv() == mock._ZN3afx25frames32_stat_tmp_doublesEv()
    for n in (1, 63, 64, 1023, 1024, 50000):
        assert real._ZN3afx25load_scan_blocks_per_fileEi(n) == mock._ZN3afx25load_scan_blocks_per_fileEi(n)
    for n in (1, 4095, 4096, 4097, 10 ** 9):
        assert real._ZN3afx15resample_blocksEl(n) == mock._ZN3afx15resample_blocksEl(n)


def test_the_bench_lines_sharded_crawl_on_eight_mock_devices(tmp_path):
    """bench.py's config.sharded_crawl as an 8-GPU node would run it -- one process, afec_amd/hostlib.py ->
    afec_crawl_wave_images_ex -> afec::TCrawler over devices 0..7 -- with the mock device under the same host sources
    (build.sh's host_lib target, injected into hostlib from the test, never reachable from the product): the per-device
    arrays the Python side sizes by the number of devices, file i -> device i mod 8, workers per device from the CPU
    quota, and the same row digest for the same content on every device."""
    out = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "build.sh"), "plain", "host_lib"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    mock = out.stdout.decode().split()[-1]
    script = tmp_path / "crawl8.py"
    script.write_text(
        "import ctypes, json, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import afec_amd.hostlib as hostlib\n"
        "L = ctypes.CDLL(sys.argv[1]); L.hipstub_set_device_count(8)\n"
        "hostlib._lib = hostlib._bind(L)\n"
        "import bench\n"
        "bench.afx.device_count = lambda: 8\n"
        "print(json.dumps(bench.sharded_crawl(8, 150, 99, repeats=2)))\n")
    import json
    import sys
    r = subprocess.run([sys.executable, str(script), mock], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    sc = json.loads(r.stdout.decode().splitlines()[-1])
    assert sc["devices"] == list(range(8)) and sc["devices_visible"] == 8
    assert sc["files"] == 1200 and sc["failed"] == 0 and sc["files_per_device"] == [150] * 8
    assert len(sc["upload_GB_per_s_per_device"]) == 8 and all(v and v > 0 for v in sc["upload_GB_per_s_per_device"])
    assert 1 <= sc["workers_per_device"] <= 5 and sc["workers_per_device"] == max(1, min(5, int(sc["cpu_quota"] // 8)))
    digests = sc["row_digests"]
    assert digests["identical_per_content"] and digests["files_per_device"] == [126] * 8 and digests["rows_per_content"] == 16
