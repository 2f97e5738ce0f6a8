cd $GRAFT_REPO_ROOT/ab_base
mkdir -p ../gpurun_out
python tools/layer_profile.py 1 f16x3 > ../gpurun_out/r03_v_layers_b1.log 2>&1
cat ../gpurun_out/r03_v_layers_b1.log | grep -v amdgpu.ids
