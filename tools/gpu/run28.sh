cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
bash tools/gpu/ab.sh $O/ab AO_AMD_GEMM=direct X=1 3
bash tools/gpu/ab.sh $O/ab2 AO_AMD_GEMM=lds X=1 2
