"""A toolchain hazard of this image's hipcc (ROCm 7.2), checked on the ISA of every product kernel file.

A 64-bit constant handed to inline asm through an "s" operand -- the lane masks of afx_bands.hip's v_cndmask selects, the
lane-0 mask of afx_frames32.hip -- is materialised as `s_mov_b64 sN, <32-bit literal>` whenever the value fits a
SIGN-extended int32, while the hardware ZERO-extends the literal of a 64-bit scalar move: 0xfffffffff0000000 arrives as
0x00000000f0000000 and lanes 32..63 are lost without a diagnostic (first seen in tools/ubench/bands_rows16.hip, whose
self-check caught it; profiles/r06/isa_budget_bands_rows16.txt).  The product's masks happen to be built from two 32-bit
halves today; this test keeps it that way: no file under afec_amd/csrc may compile to a 64-bit scalar move whose literal is
a sign-extended negative (the inline constants -16..-1 are exact and allowed).

The same compile also yields the compiler's resource report: registers, scratch and waves per SIMD of every kernel are held
against tests/golden/kernel_resources.json (second test below)."""
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
    """-> (ISA text, the compiler's kernel-resource-usage remarks) of one kernel file, built with the Makefile's flags"""
    out = os.path.join(out_dir, os.path.basename(source)[:-4] + ".s")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-x", "hip", "-S",
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", out, source],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    with open(out) as f:
        return f.read(), r.stderr.decode()


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out_dir = str(tmp_path_factory.mktemp("isa"))
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(sources) >= 9
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        built = list(pool.map(lambda s: device_isa(s, out_dir), sources))
    return {os.path.basename(s): b for s, b in zip(sources, built)}


def offenders(isa):
    return [line.strip() for line in isa.splitlines() if SIGN_EXTENDED_LITERAL.match(line)]


def test_the_pattern_sees_the_hazard():
    assert offenders("\ts_mov_b64 s[12:13], 0xfffffffff0000000\n")
    assert offenders("\ts_mov_b64 s[12:13], -4096\n")
    assert offenders("\ts_and_b64 s[2:3], s[4:5], 0xffffffffe0000000\n")
    assert not offenders("\ts_mov_b64 s[2:3], -16\n\ts_mov_b64 s[4:5], 0xffffffff\n\ts_mov_b64 s[6:7], 0x1fffffff\n\ts_mov_b32 s5, 0xfffe0000\n")


def test_no_product_kernel_file_holds_a_sign_extended_64_bit_scalar_literal(compiled):
    found = {name: offenders(isa) for name, (isa, _) in compiled.items() if offenders(isa)}
    assert not found, found
    # the scan looked at real kernels: the masked selects of the band kernel are in there
    bands = compiled["afx_bands.hip"][0]
    assert bands.count("v_cndmask_b32_e64") > 50 and "s_mov_b64" in bands


def kernel_resources(remarks):
    """{demangled kernel name: {"vgprs", "scratch", "occupancy", "lds"}} from -Rpass-analysis=kernel-resource-usage"""
    out, cur = {}, None
    fields = (("vgprs", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
              ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        for key, pattern in fields:
            m = re.search(pattern, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = list(out)
    demangled = subprocess.run(["c++filt"], input="\n".join(names).encode(), stdout=subprocess.PIPE).stdout.decode().splitlines()
    short = [re.sub(r"\(.*", "", d.replace("afx::(anonymous namespace)::", "").replace("void ", "")) for d in demangled]
    return {s: out[n] for s, n in zip(short, names)}


def test_no_kernel_spills_more_or_holds_fewer_waves_than_recorded(compiled):
    """The frame, band and pitch kernels sit at 254-256 VGPRs and two waves per SIMD by design: a change that
    pushes one of them over the edge shows up as scratch, not as an error, and costs tens of per cent (the ablation builds of
    profiles/r06/isa_budget_bands_rows16.txt spilled 224-672 bytes and ran slower with a third of their instructions removed).
    tests/golden/kernel_resources.json is what the shipped build compiles to (tools/kernel_resources.py writes it); a kernel
    may get better than recorded, not worse, and no kernel may appear or vanish without the record being rewritten."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kernel_resources.json")) as f:
        recorded = json.load(f)["kernels"]
    now = {}
    for name, (_, remarks) in compiled.items():
        now.update(kernel_resources(remarks))
    assert sorted(now) == sorted(recorded), (sorted(set(now) ^ set(recorded)))
    worse = {k: (now[k], recorded[k]) for k in now
             if now[k]["scratch"] > recorded[k]["scratch"] or now[k]["occupancy"] < recorded[k]["occupancy"]}
    assert not worse, worse
    assert now["frames32_kernel<0, false>"]["scratch"] == 0 and now["frames32_kernel<0, false>"]["occupancy"] == 2   # the headline kernel
