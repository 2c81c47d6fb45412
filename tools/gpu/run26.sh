cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
python -m pytest tests/test_gpu_gva_stages.py tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_riders.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --steps 6 --warmup 2 > $O/trace.log 2>&1
python - $O/trace <<'PY' > $O/stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('logits_bwd','attention_bwd_point','aggregate_bwd_gv','logits_fwd','softmax_point')): print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
rm -rf $O/trace
cat $O/stats.txt
