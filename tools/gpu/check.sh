# GPU box: the dense / block / model tests, three bench runs (fp32, +bf16), and the per-kernel stats of a short trace.  usage: bash tools/gpu/check.sh [out dir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/check}; mkdir -p $O
python -m pytest tests/test_gpu_gva_stages.py tests/test_gpu_dense.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_riders.py tests/test_gpu_bf16.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop --dtype bf16 2>/dev/null | tail -1 | python -c "import json,sys; print('bf16', json.loads(sys.stdin.read())['ms_per_step'])"
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop --steps 6 --warmup 2 > $O/trace.log 2>&1
python - $O/trace <<'PY' > $O/stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:45]: print("%-72s %5s %10.1f %8.1f"%(r['Name'][:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/8))
PY
rm -rf $O/trace
cat $O/stats.txt
