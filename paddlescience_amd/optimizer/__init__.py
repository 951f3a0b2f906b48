from . import lr_scheduler  # noqa: F401
from .optimizer import (LBFGS, SGD, Adam, AdamW, ClipGradByGlobalNorm, ClipGradByNorm, ClipGradByValue, Momentum,  # noqa: F401
                        OptimizerList, RMSProp)

__all__ = ["Adam", "AdamW", "SGD", "Momentum", "RMSProp", "LBFGS", "OptimizerList", "lr_scheduler", "build_optimizer",
           "build_lr_scheduler"]


def build_lr_scheduler(cfg, epochs, iters_per_epoch):
    cfg = dict(cfg)
    cls = cfg.pop("name")
    cfg.update({"epochs": epochs, "iters_per_epoch": iters_per_epoch})
    return getattr(lr_scheduler, cls)(**cfg)()


def build_optimizer(cfg, model_list, epochs, iters_per_epoch):
    cfg = dict(cfg)
    lr_cfg = cfg.pop("lr")
    sch = None
    if isinstance(lr_cfg, float):
        lr = lr_cfg
    else:
        sch = lr = build_lr_scheduler(lr_cfg, epochs, iters_per_epoch)
    cls = cfg.pop("name")
    opt = {"Adam": Adam, "AdamW": AdamW, "SGD": SGD, "Momentum": Momentum, "RMSProp": RMSProp, "LBFGS": LBFGS}[cls](
        learning_rate=lr, **cfg)(model_list)
    return opt, sch
