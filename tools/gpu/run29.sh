cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
bash tools/gpu/ab.sh $O/a AO_AMD_SOFTMAX_WGS=2048 AO_AMD_SOFTMAX_WGS=1024 2
bash tools/gpu/ab.sh $O/b AO_AMD_SOFTMAX6=point X=1 2
bash tools/gpu/ab.sh $O/c AO_AMD_SOFTMAX_WGS=4096 AO_AMD_SOFTMAX_WGS=1536 2
