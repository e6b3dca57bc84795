"""META_ARCH_REGISTRY / build_model — detectron2/modeling/meta_arch/build.py:6-23."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    """Build the meta-architecture named by cfg.MODEL.META_ARCHITECTURE and move it to
    cfg.MODEL.DEVICE.  No weights are loaded."""
    meta_arch = cfg.MODEL.META_ARCHITECTURE
    model = META_ARCH_REGISTRY.get(meta_arch)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
