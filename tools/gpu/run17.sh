cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_host_rows.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03q/bench_fp32.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["loss"]); print({k:v for k,v in d["ops"].items() if not isinstance(v,str)})
PY
python bench_ops.py > $O/bench_ops.txt 2>&1; tail -8 $O/bench_ops.txt
