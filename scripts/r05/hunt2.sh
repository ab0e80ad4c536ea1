cd $GRAFT_REPO_ROOT
for mode in "NERO_DBG_SYNC=1" "NERO_DBG_SIDE_STREAM=1" "X=1"; do for i in 1 2 3 4 5 6; do echo "$mode: $(env $mode NERO_STREAMS=3 NERO_DW_JOIN=j python scripts/r05/dbg_streams.py bear 512 2>&1 | grep -c identical) of 9"; done; done
