#!/bin/bash
# round 5, re-entry: call 37 showed a 512 KB footprint costing +1.25 ms and +11 GB of fabric reads when its lines lie 4 KB apart (set conflicts).
# (e) of profiles/r05_item_pass_split.txt tested S = 1040 (rows of D 4,160 bytes apart) only at 163 user blocks, where the D window does not
# fit the L2 anyway.  Here: S = 1024 / 1040 at 163 / 200 / 256 user blocks - does the unaligned row stride pay once the window fits?
set -o pipefail
O=gpurun_out
for S in 1024 1040; do for UC in 163 200 256; do
  TMF_USER_CHUNKS=$UC timeout -k 10 300 python bench.py --no-extras --steps 10 --warmup 3 --samples $S > $O/c4ab.json 2>$O/c4ab.err || { echo "run S=$S UC=$UC failed"; tail -5 $O/c4ab.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/c4ab.json').read().strip().splitlines()[-1])
print('S=$S user_blocks=$UC', round(d['ms_per_step'],2), {k[5:]:v[0] for k,v in d['roofline']['kernels_ms'].items()}, flush=True)"
done; done 2>&1 | tee $O/r05_call38_ab.txt
