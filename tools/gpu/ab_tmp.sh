cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gva_stages.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py -m gpu -x -q -k "not equal_steps" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats -d gpurun_out/r03ao/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --steps 6 --warmup 2 > gpurun_out/r03ao/trace.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03ao/trace/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'fold' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
rm -rf gpurun_out/r03ao/trace
