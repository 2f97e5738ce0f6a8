cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "decoder_stage or gaze_head" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "golden or unusual" 2>&1 | tail -2
bash tools/decoder_prof.sh f16x3 2>&1 | grep -v amdgpu.ids
