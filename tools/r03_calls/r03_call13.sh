cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "decoder_stage" 2>&1 | tail -1
bash tools/decoder_prof.sh f16x3 2>&1 | grep -E "decoder only|dynconv"
