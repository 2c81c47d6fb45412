bash tools/gpu/ab_lib.sh gpurun_out/r03al 3
bash tools/gpu/ab.sh gpurun_out/r03al/x X=1 AO_AMD_WGRAD_XCD=0 3
