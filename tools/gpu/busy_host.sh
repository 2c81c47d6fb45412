# usage: bash tools/gpu/busy_host.sh TAG "BURNERS..." [ORDER]  -- the bench line under a busy host: B busy-loop processes (`yes`) run
# beside it (started and killed by PID here), for each B in BURNERS; graph / eager issue alternately as in fresh_box.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; BURN=${2:-"0 256"}; ORDER=${3:-"g e"}
mkdir -p gpurun_out/busy
for B in $BURN; do
  pids=""
  for j in $(seq 1 $B); do yes > /dev/null & pids="$pids $!"; done
  sleep 1
  for m in $ORDER; do
    if [ $m = g ]; then E="AO_AMD_GRAPH=1"; else E="AO_AMD_GRAPH=0"; fi
    env $E $EXTRA_ENV python bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop > gpurun_out/busy/${T}_b${B}_${m}.json 2> gpurun_out/busy/${T}_b${B}_${m}.err
    python - gpurun_out/busy/${T}_b${B}_${m}.json $m $B <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d["host"]
    print("burners %s %s ms %.3f  step min/med/p90/max %.2f %.2f %.2f %.2f  host issue %.2f cpu %.2f  load %.1f" % (
        sys.argv[3], sys.argv[2], d["ms_per_step"], h["step_ms"]["min"], h["step_ms"]["median"], h["step_ms"]["p90"], h["step_ms"]["max"],
        h["host_issue_ms"], h["host_cpu_ms"], h["loadavg"][0]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
  for p in $pids; do kill $p 2>/dev/null; done
  wait 2>/dev/null
done
