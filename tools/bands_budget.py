#!/usr/bin/env python3
"""Per-section instruction budget of bands_kernel<15>: a copy of afec_amd/csrc/afx_bands.hip gets a scheduling barrier and
an assembly comment at every section boundary of the frame loop, is compiled to ISA (hipcc -S --cuda-device-only), and the
instructions between the markers are counted by kind.  (The scheduler still moves some work across the markers; the 'park'
section holds the closed forms that run once per four frames.)   usage: tools/bands_budget.py"""
import os
os.chdir(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "afec_amd", "csrc"))
import re,subprocess,sys
src=sys.argv[1] if len(sys.argv)>1 else 'afx_bands.hip'
s=open(src).read()
def mark(after, name, s, before=False):
    assert after in s, after
    m='__builtin_amdgcn_sched_barrier(0); asm volatile("; M_%s");\n' % name
    return s.replace(after, (m+after) if before else (after+'\n'+m), 1)
s=mark('    const int64_t f = (int64_t)ch.frame0 + fi;','top',s)
s=mark('    // ---- spectrum bands 0..25 for the half-wave frame kernel','spectrum',s,True)
s=mark('    // ---- the raw sums of the spectral statistics over bins 1..738','stats',s,True)
s=mark('      const double total = read_lane<0>(red);','rolloff',s,True)
s=mark('    // ---- spectral_flux: Pearson r with the previous frame','flux',s,True)
s=mark('    // ---- masked per-band sums: lane L ends up with band','bandsums3',s,True)
s=mark('    double lg[8];','logs',s,True)
s=mark('    const double bmax = band_max(','bmax',s,True)
s=mark('    // ---- complexity: strict local maxima above','peaks',s,True)
s=mark('    // ---- contrast: sort (band, value) keys','sortprep',s,True)
s=mark('    sort_level<256>(key, lane_v);','sort',s,True)
s=mark('    sort_level<256>(key, lane_v);','cuts',s)
s=mark('    double vsum, psum;','vpsum',s,True)
s=mark('    // cuts inside a tie class: exact resolution','ties',s,True)
s=mark('    // ---- park the frame','park',s,True)
s=mark('    // this frame is the next one','tail',s,True)
open('/tmp/bands_marked.hip','w').write(s)
subprocess.check_call('/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I. -S --cuda-device-only -x hip /tmp/bands_marked.hip -o /tmp/bands_marked.s'.split(), stderr=subprocess.DEVNULL)
t=open('/tmp/bands_marked.s').read()
m=re.search(r'_ZN3afx12_GLOBAL__N_112bands_kernelILi15EEEvNS_8BandArgsE:(.*?)\.Lfunc_end', t, re.S)
body=m.group(1).split('\n')
cur=None; counts={}; order=[]
for l in body:
    x=l.strip()
    mm=re.match(r'; M_(\w+)',x)
    if mm:
        cur=mm.group(1)
        if cur not in counts: counts[cur]={'valu':0,'salu':0,'ds':0,'vmem':0}; order.append(cur)
        continue
    if cur is None or not x or x.startswith((';','.','s_waitcnt','s_nop')) or x.endswith(':'): continue
    op=x.split()[0]
    c=counts[cur]
    if op.startswith('v_'): c['valu']+=1
    elif op.startswith('s_'): c['salu']+=1
    elif op.startswith('ds_'): c['ds']+=1
    else: c['vmem']+=1
tot=0
for k in order:
    print(f"{k:12s} {counts[k]}"); tot+=counts[k]['valu']
print('total valu (park counted once)',tot)
