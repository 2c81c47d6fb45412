# usage: bash tools/gpu/r05_trace.sh OUT name  -- kernel trace of the bench on ao_amd/lib/libptv2_<name>.so: per-kernel summary, one step in launch order,
# per-(kernel, grid) durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; mkdir -p $O
cp ao_amd/lib/libptv2_$2.so ao_amd/lib/libptv2_hip.so
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop --steps 20 --warmup 5 > $O/trace.log 2>&1
python tools/prof_summary.py $O/trace 25 70 > $O/kernel_summary.txt 2>&1
python tools/trace_step.py $O/trace 12 > $O/step_sequence.txt 2>&1
python tools/trace_grids.py $O/trace 25 > $O/trace_grids.txt 2>&1
rm -rf $O/trace
head -3 $O/kernel_summary.txt; tail -1 $O/step_sequence.txt
