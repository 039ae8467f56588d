#!/usr/bin/env python3
"""Per-section instruction budget of a kernel's frame loop: a copy of the kernel's source gets a scheduling barrier and an
assembly comment at every section boundary, is compiled to ISA (hipcc -S --cuda-device-only), and the instructions between
the markers are counted by kind.  (The scheduler still moves some work across the markers.)

usage: tools/section_budget.py bands|pitch"""
import collections
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "afec_amd", "csrc")
KERNELS = {
    # name: (source, start of the kernel in the source (markers are placed behind it), mangled-name regex, markers);
    # marker = (text to find, section name, before?)
    "bands": ("afx_bands.hip", "void bands_kernel(const BandArgs a)", r"_ZN3afx12_GLOBAL__N_112bands_kernelILi15EEEvNS_8BandArgsE", [
        ("    const int64_t f = (int64_t)ch.frame0 + fi;", "top", False),
        ("    // ---- spectrum bands 0..25 for the half-wave frame kernel", "spectrum", True),
        ("    // ---- the raw sums of the spectral statistics over bins 1..738", "stats", True),
        ("      const double total = read_lane<0>(red);", "rolloff", True),
        ("    // ---- spectral_flux: Pearson r with the previous frame", "flux", True),
        ("    // ---- masked per-band sums: lane L ends up with band", "bandsums3", True),
        ("    double lg[8];", "logs", True),
        ("    const double bmax = band_max(", "bmax", True),
        ("    // ---- complexity: strict local maxima above", "peaks", True),
        ("    // ---- contrast: sort (band, value) keys", "sortprep", True),
        ("    sort_level<256>(key, lane_v);", "sort", True),
        ("    sort_level<256>(key, lane_v);", "cuts", False),
        ("    double vsum, psum;", "vpsum", True),
        ("    // cuts inside a tie class: exact resolution", "ties", True),
        ("    // ---- park the frame", "park (closed forms: once per four frames)", True),
        ("    // this frame is the next one", "tail", True)]),
    "pitch": ("afx_time.hip", "void pitch_kernel(const TimeArgs a)", r"_ZN3afx12_GLOBAL__N_112pitch_kernelIfLb1EEEvNS_8TimeArgsE", [
        ("      cx<double> zn[16];", "convert + prefetch", True),
        ("      fft(zn, c);", "forward transform", True),
        ("      const double sign = (lane & 1) ? -1.0 : 1.0;", "pair-wise spectral product", True),
        ("      // the second half's transform is the next frame's first-half transform", "hand-over + blocked loads", True),
        ("      fft(g, c);", "inverse transform", True),
        ("      // c[2m] = Re F[m] / 1024", "to the blocked layout", True),
        ("      // ---- squared-difference terms (pitchyinfast.c:96-117) ----", "hop descriptors + squares", True),
        ("      double s0 = 0.0, s1 = 0.0;", "squared differences", True),
        ("      // ---- cumulative mean normalisation", "cumulative mean normalisation", True),
        ("      // ---- first dip below the tolerance", "minimum search", True),
        ("      // fvec_quadratic_peak_pos", "interpolation + confidence", True)]),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bands"
    source, start, mangled, markers = KERNELS[name]
    text = open(os.path.join(CSRC, source)).read()
    at = text.index(start)
    head, body = text[:at], text[at:]
    for find, section, before in markers:
        assert find in body, find
        mark = '__builtin_amdgcn_sched_barrier(0); asm volatile("; M_%s");\n' % re.sub(r"\W+", "_", section)
        body = body.replace(find, (mark + find) if before else (find + "\n" + mark), 1)
    marked = f"/tmp/{name}_marked.hip"
    open(marked, "w").write(head + body)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(CSRC, "..", "..", "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", "-x", "hip", marked, "-o", marked[:-4] + ".s"], stderr=subprocess.DEVNULL)
    isa = open(marked[:-4] + ".s").read()
    lines = re.search(mangled + r":(.*?)\.Lfunc_end", isa, re.S).group(1).split("\n")
    cur, counts, ops, order = None, {}, collections.defaultdict(collections.Counter), []
    for line in lines:
        x = line.strip()
        m = re.match(r"; M_(\w+)", x)
        if m:
            cur = m.group(1)
            if cur not in counts:
                counts[cur] = collections.Counter()
                order.append(cur)
            continue
        if cur is None or not x or x.startswith((";", ".", "s_waitcnt", "s_nop")) or x.endswith(":"):
            continue
        op = x.split()[0]
        ops[cur][op] += 1
        counts[cur]["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "ds" if op.startswith("ds_") else "vmem"] += 1
    total = 0
    print(f"# {name}: VALU / scalar / DS / memory instructions per section of the frame loop ({source}, compiler's ISA, tools/section_budget.py)")
    for k in order:
        c = counts[k]
        total += c["valu"]
        print(f"{k:46s} valu {c['valu']:5d}  salu {c['salu']:4d}  ds {c['ds']:4d}  vmem {c['vmem']:3d}   most: " +
              ", ".join(f"{o} {n}" for o, n in ops[k].most_common(4)))
    print(f"total VALU {total}")


if __name__ == "__main__":
    main()
