"""META_ARCH_REGISTRY and build_model(cfg) — detectron2/modeling/meta_arch/build.py:6-23: look the class
named by cfg.MODEL.META_ARCHITECTURE up, construct it from cfg, move it to cfg.MODEL.DEVICE (no weights)."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    arch_cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    return arch_cls(cfg).to(torch.device(cfg.MODEL.DEVICE))
