"""Registration of this package's classes under the reference's registries, with the reference's own idiom.

The reference builds its model, segmentor and optimizer from config dicts through `Registry` objects
(pointcept/utils/registry.py:59-): `MODELS` (pointcept/models/builder.py), `OPTIMIZERS` (pointcept/utils/optimizer.py:12-17,
`build_optimizer` ends in `OPTIMIZERS.build(cfg=cfg)`, :55).  `register(MODELS=..., OPTIMIZERS=...)` is what the one file a
maintainer adds calls:

    from pointcept.models.builder import MODELS
    from pointcept.utils.optimizer import OPTIMIZERS
    import ao_amd.ptv2.registry as mi355x
    mi355x.register(MODELS=MODELS, OPTIMIZERS=OPTIMIZERS)

after which `model = dict(type="DefaultSegmentor", backbone=dict(type="PT-v2m2", ...))` resolves to the MI355X classes and
`optimizer = dict(type="FlatAdamW", lr=0.006, weight_decay=0.05)` is a CONFIG edit (the reference's line is
`dict(type="AdamW", lr=0.006, weight_decay=0.05)`, configs/s3dis/semseg-pt-v2m2-0-base.py:42): no trainer code changes.
`FlatAdamW.step()` takes the `.grad`s autograd (or DistributedDataParallel) delivered; when they alias the native backward's
flat gradient buffer (the default delivery of ao_amd/ptv2/native_model.py) no copy is made.
"""


def _register(registry, cls, name, force):
    """`registry.register_module(module=cls, name=name)` -- pointcept/utils/optimizer.py:15-17 -- tolerant of a name that is
    already taken when `force` (the reference's Registry takes force=True to replace an entry, registry.py `_register_module`)."""
    try:
        registry.register_module(module=cls, name=name, force=force)
    except TypeError:  # a registry without the force keyword
        registry.register_module(module=cls, name=name)


def register(MODELS=None, OPTIMIZERS=None, force=True):
    """Put PT-v2m2, the two segmentors and FlatAdamW under the reference's registry names.  Either registry may be omitted."""
    done = []
    if MODELS is not None:
        from .model import PointTransformerV2
        from .segmentor import DefaultSegmentor, DefaultSegmentorSAM_Image

        for cls, name in ((PointTransformerV2, "PT-v2m2"), (DefaultSegmentor, "DefaultSegmentor"),
                          (DefaultSegmentorSAM_Image, "DefaultSegmentorSAM_Image")):
            _register(MODELS, cls, name, force)
            done.append(name)
    if OPTIMIZERS is not None:
        from .optim import FlatAdamW

        _register(OPTIMIZERS, FlatAdamW, "FlatAdamW", force)
        done.append("FlatAdamW")
    return done
