# usage: bash tools/gpu/r05_kernel_ab.sh OUT "grep pattern" name1 name2 ...  -- kernel trace of the bench on each ao_amd/lib/libptv2_<name>.so,
# the summary lines that match the pattern side by side (per-launch averages of one kernel family across builds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; P="$2"; shift 2; mkdir -p $O
for t in "$@"; do
  cp ao_amd/lib/libptv2_$t.so ao_amd/lib/libptv2_hip.so
  rocprofv3 --kernel-trace --stats -d $O/trace_$t -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop --steps 20 --warmup 5 > $O/trace_$t.log 2>&1
  python tools/prof_summary.py $O/trace_$t 25 70 > $O/kernel_summary_$t.txt 2>&1
  python tools/trace_step.py $O/trace_$t 12 > $O/step_sequence_$t.txt 2>&1
  rm -rf $O/trace_$t
  echo "== $t: $(tail -1 $O/step_sequence_$t.txt)"
  grep -E "$P" $O/kernel_summary_$t.txt | cut -c1-100,150-200
done
cp ao_amd/lib/libptv2_$1.so ao_amd/lib/libptv2_hip.so
