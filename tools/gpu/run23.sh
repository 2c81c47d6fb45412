cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
python -m pytest tests/test_gpu_block.py tests/test_gpu_gva_stages.py tests/test_gpu_riders.py tests/test_gpu_model.py tests/test_gpu_native_model.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/gpu/ab.sh $O/ab X=1 AO_AMD_LOGITS_BWD6=0 3
bash tools/gpu/ab.sh $O/ab2 AO_AMD_LOGITS_BWD6_WGS=512 AO_AMD_LOGITS_BWD6_WGS=768 2
