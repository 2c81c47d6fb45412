cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --no-cpu-baseline --no-ops > $O/bench_scannet.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --scenes 4 --points 80000 --no-cpu-baseline --no-ops > $O/bench_4x80k.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_bf16","bench_scannet","bench_4x80k"):
    try:
        d=json.loads(open("gpurun_out/r03m/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f,"FAILED",e)
PY
B="python3 bench.py --no-cpu-baseline --no-ops --no-roofline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $O/pmc_sq -- $B --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1
python tools/pmc_sq.py $O/pmc_sq > $O/sq_counters.jsonl 2> $O/sq.err
rm -rf $O/pmc_sq
head -c 3000 $O/sq_counters.jsonl
