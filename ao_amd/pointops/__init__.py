"""Drop-in for the reference's `pointops` package (libs/pointops/functions/__init__.py:1-14),
backed by libptv2_hip.so.  Same names, argument order, dtypes and placeholder conventions."""
from .query import knn_query, knn_query_dist2, ball_query, random_ball_query
from .sampling import farthest_point_sampling
from .grouping import grouping, grouping2
from .interpolation import interpolation, interpolation2
from .subtraction import subtraction
from .aggregation import aggregation
from .attention import attention_relation_step, attention_fusion_step
from .utils import (
    query_and_group,
    knn_query_and_group,
    ball_query_and_group,
    batch2offset,
    offset2batch,
)
