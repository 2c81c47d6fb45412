# usage: bash tools/gpu/ab_multi.sh OUT reps name1 name2 ...   -- alternates ao_amd/lib/libptv2_<name>.so (built by hand) under the same
# bench, `reps` rounds over all of them; the library named first is left in place at the end
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; R=$2; shift 2; mkdir -p $O
L=ao_amd/lib
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop > /dev/null 2>&1
for i in $(seq 1 $R); do
  for t in "$@"; do
    cp $L/libptv2_$t.so $L/libptv2_hip.so
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop 2>/dev/null | tail -1 | python -c "import json,sys; print('$t', json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp $L/libptv2_$1.so $L/libptv2_hip.so
