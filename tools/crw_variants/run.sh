#!/bin/bash
# GPU box: time the cr_wide_kernel variants on the n = 56 systems of tools/wide_rate.py
cd "$(dirname "$0")"
python3 - <<'PY'
import sys; sys.path.insert(0, "../..")
import numpy as np
from geconpy_amd import workloads as wl
n, ns, nl = 56, 25, 16
sysm = [wl.sw_shaped_system(7000 + 5 * n + i, n=n, n_state=ns, n_lead=nl, k=7) for i in range(64)]
np.concatenate([np.stack([s_[j] for s_ in sysm]).ravel() for j in range(3)]).tofile("/tmp/crw_in.bin")
PY
for v in "$@"; do ./crw_v$v /tmp/crw_in.bin 56; done
