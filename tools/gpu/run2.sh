cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
python -m pytest tests/test_gpu_block.py tests/test_gpu_gva_stages.py tests/test_gpu_model.py -m gpu -x -q -k "not equal_steps" > gpurun_out/r03b/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03b/pytest.log
tail -5 gpurun_out/r03b/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops > gpurun_out/r03b/bench_fp32.json 2> gpurun_out/r03b/bench_fp32.err
AO_AMD_LOGITS_BWD=staged python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops > gpurun_out/r03b/bench_staged.json 2> gpurun_out/r03b/bench_staged.err
python - <<'PY'
import json
for f in ("bench_fp32","bench_staged"):
    try:
        d=json.loads(open("gpurun_out/r03b/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"]["loss"])
        ak=d["roofline"]["all_kernels"]
        for k in ("logits_bwd_params_kernel","logits_bwd_rows_kernel","logits_bwd_gather_kernel"):
            if k in ak: print("   ",k,ak[k])
    except Exception as e: print(f,"FAILED",e)
PY
