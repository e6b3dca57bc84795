from .config import CfgNode, get_cfg
from .defaults import add_wsl_config

__all__ = ["CfgNode", "get_cfg", "add_wsl_config"]
