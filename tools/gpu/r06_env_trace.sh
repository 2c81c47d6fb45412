# usage: bash tools/gpu/r06_env_trace.sh OUT "grep pattern" "ENV_A" "ENV_B" ...  -- kernel trace of the bench under each environment (same library),
# the summary lines that match the pattern side by side
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; P="$2"; shift 2; mkdir -p $O
i=0
for e in "$@"; do
  i=$((i+1)); t=env$i
  export $e
  rocprofv3 --kernel-trace --stats -d $O/trace_$t -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop --steps 20 --warmup 5 > $O/trace_$t.log 2>&1
  unset ${e%%=*}
  python tools/prof_summary.py $O/trace_$t 25 90 > $O/kernel_summary_$t.txt 2>&1
  python tools/trace_step.py $O/trace_$t 12 > $O/step_sequence_$t.txt 2>&1
  rm -rf $O/trace_$t
  echo "== $e: $(tail -1 $O/step_sequence_$t.txt)"
  grep -E "$P" $O/kernel_summary_$t.txt | cut -c1-70,93-150
done
