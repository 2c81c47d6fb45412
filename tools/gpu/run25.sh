cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
python -m pytest tests/test_gpu_block.py -m gpu -x -q -k "fused_logits" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/gpu/ab.sh $O/ab X=1 AO_AMD_LOGITS_BWD6=0 3
