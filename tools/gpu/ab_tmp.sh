cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stage_api.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_rccl.py tests/test_gpu_bf16.py -m gpu -x -q -k "not equal_steps" 2>&1 | tail -3
bash tools/gpu/ab.sh gpurun_out/r03aq/x X=1 AO_AMD_WGRAD_DEFER=0 4
