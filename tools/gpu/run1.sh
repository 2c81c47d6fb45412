cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_fp32.json 2> gpurun_out/r03a/bench_fp32.err
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
head -c 1500 gpurun_out/r03a/bench_fp32.json
