# usage: bash tools/gpu/r05_trace_env.sh OUT "ENV..."  -- kernel trace of the bench under an environment: one step in launch order with the queues,
# the gaps of the compute queue inside that step, per-kernel summary
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
for e in "$@"; do export $e; done
rocprofv3 --kernel-trace --stats -d $O/trace -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop --steps 20 --warmup 5 > $O/trace.log 2>&1
python tools/trace_step.py $O/trace 12 > $O/step_sequence.txt 2>&1
python tools/trace_gaps.py $O/trace > $O/trace_gaps.txt 2>&1
python - $O/step_sequence.txt <<'PY' > $O/queue_gaps.txt
import sys
rows=[l.split(None,4) for l in open(sys.argv[1]) if l[:1]==' ' or l[:1].isdigit()]
rows=[(float(r[0]),float(r[1]),r[2],r[4].strip()) for r in rows if len(r)>=5 and r[2].startswith('q')]
for q in sorted(set(r[2] for r in rows)):
    rs=[r for r in rows if r[2]==q]; end=None; tot=0; big=[]
    for s,d,_,name in rs:
        if end is not None and s-end>3: tot+=s-end; big.append((round(s-end,1), round(s,1), name[:50]))
        end=max(end or 0, s+d)
    print(q, "launches", len(rs), "busy %.1f us"%sum(r[1] for r in rs), "gaps>3us total %.1f us"%tot)
    for g in sorted(big, reverse=True)[:12]: print("   gap %7.1f us before t=%9.1f %s"%g)
PY
rm -rf $O/trace
cat $O/queue_gaps.txt; tail -1 $O/step_sequence.txt
