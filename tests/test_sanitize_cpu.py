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
