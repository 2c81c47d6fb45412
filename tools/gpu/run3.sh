cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o /tmp/anyorder_probe > $O/probe_build.log 2>&1 && timeout 120 /tmp/anyorder_probe > $O/anyorder_probe.txt 2>&1
cat $O/anyorder_probe.txt
python -m pytest tests/test_gpu_block.py tests/test_gpu_gva_stages.py tests/test_gpu_model.py tests/test_gpu_native_model.py -m gpu -x -q -k "not equal_steps" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops > $O/bench_fp32.json 2> $O/bench_fp32.err
AO_AMD_PEB=flat python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops > $O/bench_pebflat.json 2> $O/bench_pebflat.err
python - <<'PY'
import json
for f in ("bench_fp32","bench_pebflat"):
    try:
        d=json.loads(open("gpurun_out/r03c/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"]["loss"])
        ak=d["roofline"]["all_kernels"]
        for k in ("peb_fwd_kernel","bn_stats_kernel"):
            if k in ak: print("   ",k,ak[k])
    except Exception as e: print(f,"FAILED",e)
PY
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --steps 12 --warmup 3 > $O/trace.log 2>&1
python tools/trace_step.py $O/trace 8 > $O/step_sequence.txt 2>&1
python tools/prof_summary.py $O/trace 15 70 > $O/kernel_summary.txt 2>&1
rm -rf $O/trace
tail -3 $O/step_sequence.txt
