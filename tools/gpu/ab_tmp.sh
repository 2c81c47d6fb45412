cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aj; mkdir -p $O
python -m pytest tests/test_gpu_gva_stages.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_ops.py -m gpu -x -q -k "not equal_steps" 2>&1 | tail -3
bash tools/gpu/ab.sh $O/a X=1 AO_AMD_AGG_SOFTMAX=split 4
