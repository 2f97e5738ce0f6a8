cd $GRAFT_REPO_ROOT
timeout 300 python tools/bneck_after_mfma_probe.py 2>&1 | grep -v amdgpu.ids
