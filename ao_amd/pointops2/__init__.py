"""pointops2-style spellings (libs/pointops2/functions/pointops.py:15-55,963-1001,1023-1192) of the
same HIP ops.  Only the six base ops the two libraries share are provided; the
Stratified-Transformer attention / rpe kernels are out of scope."""
from . import pointops  # noqa: F401
