# usage: bash tools/gpu/ab_lib.sh OUT reps   -- alternates ao_amd/lib/libptv2_old.so / libptv2_new.so (built by hand) under the same bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; R=${2:-3}; mkdir -p $O
L=ao_amd/lib
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop > /dev/null 2>&1
for i in $(seq 1 $R); do
  for t in old new; do
    cp $L/libptv2_$t.so $L/libptv2_hip.so
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop 2>/dev/null | tail -1 | python -c "import json,sys; print('$t', json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp $L/libptv2_new.so $L/libptv2_hip.so
