#!/bin/bash
# round 4: C3 / C4 / C4-everything profiles of the current build
set -u
export AFX_ROUND=r04
AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3 --workload c3 --mask frame | head -14
python tools/profile_config.py c4 --workload c4 --mask frame | head -14
python tools/profile_config.py c4_everything --workload c4 --mask everything | head -20
