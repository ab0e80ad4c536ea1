cd $GRAFT_REPO_ROOT
for j in l e; do for i in 1 2 3 4 5 6 7 8; do echo "join=$j: $(NERO_STREAMS=3 NERO_DW_JOIN=$j python scripts/r05/dbg_streams.py bear 512 2>&1 | grep -c identical) of 9"; done; done
for i in 1 2 3 4; do echo "bell 2048 join=l: $(NERO_STREAMS=3 NERO_DW_JOIN=l python scripts/r05/dbg_streams.py bell 2048 2>&1 | grep -c identical) of 9"; done
