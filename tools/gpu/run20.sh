cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
for s in 0 1 2 3; do DIAG_SEED=$s python tools/gpu/diag_outlier.py X=1 AO_AMD_MODEL=python 2>&1 | grep -v amdgpu.ids; done > $O/diag2.txt
DIAG_POINTS=20000 python tools/gpu/diag_outlier.py X=1 AO_AMD_MODEL=python 2>&1 | grep -v amdgpu.ids >> $O/diag2.txt
cut -c1-250 $O/diag2.txt
