cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_block.py tests/test_gpu_dense.py -m gpu -x -q -k "wgrad or linear" 2>&1 | tail -2
bash tools/gpu/ab.sh gpurun_out/r03am/x AO_AMD_WG_ODD=1 AO_AMD_WG_ODD=0 4
