cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03_n_gputest.log 2>&1
tail -3 gpurun_out/r03_n_gputest.log; grep "backbone only" gpurun_out/r03_n_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
