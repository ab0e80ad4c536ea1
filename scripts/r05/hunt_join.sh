#!/bin/bash
# race hunt (round 5): where may the weight-gradient stream be joined without losing bit reproducibility?  scripts/r05/hunt_join.sh "<modes>" [runs]
cd "$(dirname "$0")/../.."
for j in $1; do for i in $(seq 1 ${2:-6}); do echo "join=$j: $(NERO_STREAMS=3 NERO_DW_JOIN=$j python scripts/r05/dbg_streams.py bear 512 2>&1 | grep -c identical) of 9 identical"; done; done
