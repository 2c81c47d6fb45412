"""Put this directory's parent (`dropin/`) on PYTHONPATH and `import pointops` resolves to the
MI355X implementation -- pointcept model code runs unchanged."""
from ao_amd.pointops import *  # noqa: F401,F403
from ao_amd.pointops import knn_query_dist2  # noqa: F401
