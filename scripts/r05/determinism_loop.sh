#!/bin/bash
# the Stage-I determinism test repeated under stream modes: scripts/r05/determinism_loop.sh "<NERO_STREAMS values...>"
cd "$(dirname "$0")/../.."
for s in "$@"; do
  NERO_STREAMS=$s timeout 300 python -m pytest tests/test_determinism.py -q -k "stage1" --tb=line 2>&1 | grep -E "passed|failed|Error|assert" | head -3 | sed "s/^/streams=$s: /"
done
