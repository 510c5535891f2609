#!/bin/bash
# round 5: smoke() as the driver runs it, on the final tree
set -o pipefail
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
