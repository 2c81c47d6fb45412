# usage: bash tools/gpu/fresh_box.sh TAG [ORDER]   -- one call per fresh box: the default bench line (minus the CPU leg) as the FIRST
# process on the box, then eager / graph issue alternately; prints ms per step, the per-step spread and the host figures.
# ORDER = "g e g" (default) or e.g. "e g e": g = graph issue (default build), e = AO_AMD_GRAPH=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; ORDER=${2:-"g e g"}
mkdir -p gpurun_out/fresh
i=0
for m in $ORDER; do
  i=$((i+1))
  if [ $m = g ]; then E="AO_AMD_GRAPH=1"; else E="AO_AMD_GRAPH=0"; fi
  env $E python bench.py --no-cpu-baseline --no-ops > gpurun_out/fresh/${T}_${i}${m}.json 2> gpurun_out/fresh/${T}_${i}${m}.err
  python - gpurun_out/fresh/${T}_${i}${m}.json $m <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d["host"]
    print("%s ms %.3f  step min/med/p90/max %.2f %.2f %.2f %.2f  host issue %.2f cpu %.2f  load %.1f  wgrad %.1f us" % (
        sys.argv[2], d["ms_per_step"], h["step_ms"]["min"], h["step_ms"]["median"], h["step_ms"]["p90"], h["step_ms"]["max"],
        h["host_issue_ms"], h["host_cpu_ms"], h["loadavg"][0], d["roofline"]["avg_us"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
