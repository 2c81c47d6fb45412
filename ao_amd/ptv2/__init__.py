"""State_dict-compatible PT-v2m2 ("PT-v2m2") on the MI355X ops."""
from .model import (  # noqa: F401
    Block,
    BlockSequence,
    Decoder,
    Encoder,
    GridPool,
    GroupedVectorAttention,
    GVAPatchEmbed,
    PointBatchNorm,
    PointTransformerV2,
    UnpoolWithSkip,
    build_from_cfg,
)
from .basket import LogitBasket  # noqa: F401
from .segmentor import DefaultSegmentor, DefaultSegmentorSAM_Image, S3DIS_BACKBONE, SCANNET_BACKBONE  # noqa: F401
