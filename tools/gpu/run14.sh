cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests/test_gpu_dense.py tests/test_gpu_block.py tests/test_gpu_gva_stages.py tests/test_gpu_model.py tests/test_gpu_native_model.py tests/test_gpu_rccl.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in base1 base2 bf16; do
  case $v in base*) E="";; bf16) E="--dtype bf16";; esac
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops $E > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for f in ("base1","base2","bf16"):
    try:
        d=json.loads(open("gpurun_out/r03n/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"]["loss"])
    except Exception as e: print(f,"FAILED",e)
PY
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --steps 12 --warmup 3 > $O/trace.log 2>&1
python tools/trace_step.py $O/trace 8 > $O/step_sequence.txt 2>&1
rm -rf $O/trace
tail -2 $O/step_sequence.txt
