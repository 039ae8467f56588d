"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/afx.h declares,
and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import os
import re

import pytest

import afec_amd
from afec_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.library_path()):
        capi.build_library()
    return capi.load_library()


def header_functions():
    text = open(os.path.join(ROOT, "include", "afx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(afx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), name


def test_no_oracle_or_reference_in_product():
    """The product must not route through oracle/ or /root/reference."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "afec_amd")):
        if os.sep + "lib" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "afx_oracle" not in text and "/root/reference" not in text, os.path.join(dirpath, f)
                assert "oracle/" not in text.replace("checked against oracle/", ""), os.path.join(dirpath, f)


def test_status_strings(lib):
    assert lib.afx_status_str(0) == b"ok"
    assert b"device" in lib.afx_status_str(-3)


def test_shipped_library_has_no_diagnostic_or_ablation_switches(lib):
    info = lib.afx_build_info().decode()
    assert "arch=gfx950" in info and "stamps=0" in info and "ablation=0" in info, info
    # the switches themselves must not be set by the product Makefile
    mk = open(os.path.join(ROOT, "afec_amd", "csrc", "Makefile")).read()
    product_flags = [l for l in mk.splitlines() if l.startswith("CXXFLAGS")]
    assert product_flags and all("AFX_STAMPS" not in l and "AFX_ABL" not in l for l in product_flags)


def test_plan_create_fails_loudly_without_gpu(lib):
    import ctypes
    n = ctypes.c_int(0)
    have_gpu = False
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        pass
    if have_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(afec_amd.AfxError) as ei:
        afec_amd.Plan()
    assert ei.value.status == -3


def test_unsupported_geometry_is_rejected_before_touching_the_device(lib):
    with pytest.raises(afec_amd.AfxError) as ei:
        afec_amd.Plan(sample_rate=48000)
    assert ei.value.status == -2
    with pytest.raises(afec_amd.AfxError) as ei:
        afec_amd.Plan(fft_size=1000)
    assert ei.value.status == -1


def test_loading_the_library_leaves_the_environment_alone():
    """libafx_hip.so changes nothing process-wide when it is loaded: GPU_MAX_HW_QUEUES stays unset (the hardware-queue
    wish is afec::TCrawler's, TCrawlOptions::mHardwareQueues) -- checked in a fresh process through libc's getenv
    (os.environ does not see setenv from C)."""
    import subprocess
    import sys
    so = os.path.join(os.path.dirname(afec_amd.__file__), "lib", "libafx_hip.so")
    code = ("import ctypes, sys; ctypes.CDLL(sys.argv[1]); c = ctypes.CDLL(None); c.getenv.restype = ctypes.c_char_p; "
            "v = c.getenv(b'GPU_MAX_HW_QUEUES'); print('unset' if v is None else v.decode())")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.check_output([sys.executable, "-c", code, so], env=env).decode().strip() == "unset"
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.check_output([sys.executable, "-c", code, so], env=env).decode().strip() == "4"
