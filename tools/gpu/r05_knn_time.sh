# usage: bash tools/gpu/r05_knn_time.sh name1 name2 ...  -- the k = 16 self query of the bench scene (120 k points) on each named library build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in "$@"; do
cp ao_amd/lib/libptv2_$l.so ao_amd/lib/libptv2_hip.so
python - "$l" <<'PY'
import sys, torch
from ao_amd import pointops, synth
b = synth.scene_batch([0], point_max=120000, room=1)
xyz = torch.from_numpy(b["coord"]).cuda(); off = torch.tensor([xyz.shape[0]], dtype=torch.int32).cuda()
for _ in range(5): pointops.knn_query_dist2(16, xyz, off)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): pointops.knn_query_dist2(16, xyz, off)
e1.record(); torch.cuda.synchronize(); print("%s knn k=16 self 120k: %.1f us per call" % (sys.argv[1], e0.elapsed_time(e1)*1e3/100))
PY
done
