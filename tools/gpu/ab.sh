# usage: bash tools/gpu/ab.sh OUTDIR "ENV_A" "ENV_B" [reps] [extra bench args]  -- alternating A/B benches, prints ms per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; A="$2"; B="$3"; R=${4:-3}; X="$5"
mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop $X > /dev/null 2>&1   # warm the box up
for i in $(seq 1 $R); do
  env $A python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop $X > $O/a$i.json 2> $O/a$i.err
  env $B python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop $X > $O/b$i.json 2> $O/b$i.err
done
python - $O $R <<'PY'
import json,sys
O,R=sys.argv[1],int(sys.argv[2])
for t in "ab":
    v=[]
    for i in range(1,R+1):
        try: v.append(json.loads(open("%s/%s%d.json"%(O,t,i)).read().strip().splitlines()[-1])["ms_per_step"])
        except Exception as e: v.append(float('nan'))
    print(t, " ".join("%.3f"%x for x in v), " min %.3f"%min(v))
PY
