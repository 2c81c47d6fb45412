"""ao_amd: MI355X-native Point Transformer V2 (PT-v2m2) hot path.

Layout:
  ao_amd/csrc/      hand-written HIP kernels (gfx950) + the C-ABI (include/ptv2_hip.h)
  ao_amd/_lib.py    ctypes binding of libptv2_hip.so (fails loudly when missing)
  ao_amd/pointops/  drop-in for the reference's `pointops` python API
  ao_amd/pointops2/ pointops2-style spellings of the same ops
  ao_amd/ptv2/      state_dict-compatible "PT-v2m2" backbone on the fused HIP ops
  ao_amd/synth.py   synthetic S3DIS-shaped scenes for tests and bench.py
"""
__version__ = "0.1.0"
