"""The C++ host layer (afec_amd/host: TSampleAnalyser / TSampleDescriptors mirror; it computes nothing itself).
The test program links the oracle as the checker; the host library itself only links libafx_hip."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "host_test")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "afec_amd", "csrc")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "afec_amd", "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libafx_oracle.so"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "afec_amd", "lib")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-o", BIN, os.path.join(ROOT, "tests", "host", "test_sample_analyser.cpp"),
        "-L" + lib, "-lafx_host", "-lafx_hip", "-L" + os.path.join(ROOT, "oracle"), "-lafx_oracle", "-lm",
        "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])


def run(mode):
    build()
    out = subprocess.run([BIN, mode], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_host_layer_fails_loudly_without_a_device():
    run("nodevice")


@pytest.mark.gpu
def test_host_sample_analyser_matches_oracle():
    run("analyse")
