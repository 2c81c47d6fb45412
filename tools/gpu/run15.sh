cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
for v in base1 off1 base2 off2; do
  case $v in base*) E="";; off*) E="AO_AMD_GEMM_COUNT_AWARE=0";; esac
  env $E python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for f in ("base1","off1","base2","off2"):
    try:
        d=json.loads(open("gpurun_out/r03o/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"]["loss"])
    except Exception as e: print(f,"FAILED",e)
PY
