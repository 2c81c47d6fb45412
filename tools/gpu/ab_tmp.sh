cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stage_api.py tests/test_gpu_gva_stages.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_riders.py tests/test_gpu_bf16.py tests/test_gpu_dense.py -m gpu -x -q -k "not equal_steps" 2>&1 | tail -3
bash tools/gpu/ab.sh gpurun_out/r03ap/x X=1 AO_AMD_BP2_GRAD=split 4
