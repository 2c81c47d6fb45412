cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
uptime
bash tools/gpu/ab.sh gpurun_out/r03as/a AO_AMD_ABP_NW=4 X=1 3
bash tools/gpu/ab.sh gpurun_out/r03as/b AO_AMD_ABP_NW=2 X=1 3
uptime
