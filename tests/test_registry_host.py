"""CPU: ao_amd.ptv2.registry puts this package's classes under the reference's registry names with the reference's own
`register_module` idiom, so that `optimizer = dict(type="FlatAdamW", ...)` / `backbone = dict(type="PT-v2m2", ...)` are config
edits (pointcept/utils/optimizer.py:12-17,55; pointcept/models/builder.py).  Uses the reference's Registry class where the
reference tree is present (this container), a minimal stand-in with the same two methods otherwise."""
import importlib
import os
import sys
import types

import torch

REF_REGISTRY = "/root/reference/pointcept/utils/registry.py"


def _registry_class():
    if os.path.exists(REF_REGISTRY):
        try:  # registry.py imports `.misc`: give the directory a package name of its own (pointcept itself is not importable here)
            pkg = types.ModuleType("_ref_pointcept_utils")
            pkg.__path__ = [os.path.dirname(REF_REGISTRY)]
            sys.modules.setdefault("_ref_pointcept_utils", pkg)
            return importlib.import_module("_ref_pointcept_utils.registry").Registry
        except ImportError:
            pass

    class Registry:  # register_module(name, force, module) / build(cfg): the two calls the reference makes
        def __init__(self, name):
            self.name, self._d = name, {}

        def register_module(self, name=None, force=False, module=None):
            if not force and name in self._d:
                raise KeyError(name)
            self._d[name] = module
            return module

        def get(self, key):
            return self._d.get(key)

        def build(self, cfg):
            args = dict(cfg)
            return self._d[args.pop("type")](**args)

    return Registry


def test_flat_adamw_builds_from_the_optimizer_config_line():
    from ao_amd.ptv2 import registry

    Registry = _registry_class()
    OPTIMIZERS, MODELS = Registry("optimizers"), Registry("models")
    OPTIMIZERS.register_module(module=torch.optim.AdamW, name="AdamW")  # pointcept/utils/optimizer.py:17
    done = registry.register(MODELS=MODELS, OPTIMIZERS=OPTIMIZERS)
    assert set(done) == {"PT-v2m2", "DefaultSegmentor", "DefaultSegmentorSAM_Image", "FlatAdamW"}
    assert OPTIMIZERS.get("FlatAdamW") is not None and MODELS.get("PT-v2m2") is not None
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in net.parameters()]
    # build_optimizer: cfg.params = model.parameters(); return OPTIMIZERS.build(cfg=cfg)   (utils/optimizer.py:20-22,55)
    cfg = dict(type="FlatAdamW", lr=0.006, weight_decay=0.05, params=net.parameters())
    opt = OPTIMIZERS.build(cfg)
    assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[0]["lr"] == 0.006 and opt.param_groups[0]["weight_decay"] == 0.05
    # the parameters are views of ONE flat buffer now, values unchanged; an LR scheduler drives param_groups as for AdamW
    base = opt.flat_param.data_ptr()
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p.detach(), b)
        assert base <= p.data_ptr() < base + 4 * opt.flat_param.numel()
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    assert sched.get_last_lr() == [0.006]
    # registering twice (an edited file re-imported) replaces the entry instead of raising
    registry.register(OPTIMIZERS=OPTIMIZERS)
