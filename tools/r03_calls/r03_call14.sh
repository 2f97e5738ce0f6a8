cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "decoder_stage or gaze_head or mlp_chain" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "golden or unusual or error_growth" 2>&1 | tail -3
bash tools/decoder_prof.sh f16x3 2>&1 | grep -v amdgpu.ids | head -14
