cd $GRAFT_REPO_ROOT
for a in "448 28 28 1 128 20 128 1" "448 28 28 1 0 20 128 1"; do python tools/bneck_bench.py $a 2>&1 | grep -E "bneck_x3|tile [0-3]:"; done
