"""A toolchain hazard of this image's hipcc (ROCm 7.2), checked on the ISA of every product kernel file.

A 64-bit constant handed to inline asm through an "s" operand -- the lane masks of afx_bands.hip's v_cndmask selects, the
lane-0 mask of afx_frames32.hip -- is materialised as `s_mov_b64 sN, <32-bit literal>` whenever the value fits a
SIGN-extended int32, while the hardware ZERO-extends the literal of a 64-bit scalar move: 0xfffffffff0000000 arrives as
0x00000000f0000000 and lanes 32..63 are lost without a diagnostic (first seen in tools/ubench/bands_rows16.hip, whose
self-check caught it; profiles/r06/isa_budget_bands_rows16.txt).  The product's masks happen to be built from two 32-bit
halves today; this test keeps it that way: no file under afec_amd/csrc may compile to a 64-bit scalar move whose literal is
a sign-extended negative (the inline constants -16..-1 are exact and allowed)."""
import glob
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "afec_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SIGN_EXTENDED_LITERAL = re.compile(r"^\s*s_(mov|and|or|xor|andn2|orn2|cselect)_b64\s.*?(0xffffffff[0-9a-f]{8}\b|(?<![\w.])-(1[7-9]|[2-9]\d|\d{3,})\b)")


def device_isa(source, out_dir):
    out = os.path.join(out_dir, os.path.basename(source)[:-4] + ".s")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-x", "hip", "-S",
                        "--cuda-device-only", "-o", out, source], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    with open(out) as f:
        return f.read()


def offenders(isa):
    return [line.strip() for line in isa.splitlines() if SIGN_EXTENDED_LITERAL.match(line)]


def test_the_pattern_sees_the_hazard():
    assert offenders("\ts_mov_b64 s[12:13], 0xfffffffff0000000\n")
    assert offenders("\ts_mov_b64 s[12:13], -4096\n")
    assert offenders("\ts_and_b64 s[2:3], s[4:5], 0xffffffffe0000000\n")
    assert not offenders("\ts_mov_b64 s[2:3], -16\n\ts_mov_b64 s[4:5], 0xffffffff\n\ts_mov_b64 s[6:7], 0x1fffffff\n\ts_mov_b32 s5, 0xfffe0000\n")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_product_kernel_file_holds_a_sign_extended_64_bit_scalar_literal(tmp_path):
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(sources) >= 9
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        texts = list(pool.map(lambda s: device_isa(s, str(tmp_path)), sources))
    found = {os.path.basename(s): offenders(t) for s, t in zip(sources, texts) if offenders(t)}
    assert not found, found
    # the scan looked at real kernels: the masked selects of the band kernel are in there
    bands = texts[[os.path.basename(s) for s in sources].index("afx_bands.hip")]
    assert bands.count("v_cndmask_b32_e64") > 50 and "s_mov_b64" in bands
