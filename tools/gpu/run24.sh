cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
for w in 512 768 1024 2048; do
AO_AMD_LOGITS_BWD6_WGS=$w rocprofv3 --kernel-trace --stats -d $O/trace$w -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --steps 6 --warmup 2 > $O/trace.log 2>&1
python - $O/trace$w <<'PY' > $O/stats$w.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('logits_bwd','attention_bwd_point_kernel<6','aggregate_bwd_gv')): print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
rm -rf $O/trace$w
echo "== $w"; cat $O/stats$w.txt
done
