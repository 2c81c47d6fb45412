cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
python -m pytest tests/test_gpu_dense.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_bf16.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/gpu/ab.sh $O/ab X=1 AO_AMD_GEMM=lds 3
bash tools/gpu/ab.sh $O/abb X=1 AO_AMD_GEMM=lds 2 "--dtype bf16"
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --steps 12 --warmup 3 > $O/trace.log 2>&1
python tools/trace_step.py $O/trace 8 > $O/step_sequence.txt 2>&1
rm -rf $O/trace
tail -2 $O/step_sequence.txt
