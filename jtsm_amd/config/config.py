"""CfgNode / get_cfg — the configuration front door of the reference (detectron2/config/config.py:12-113
on top of yacs/fvcore, neither of which is in the reference tree or this image): attribute-style nested
dict, `_BASE_` inheritance between YAML files, `merge_from_file`, `merge_from_list`, freeze/defrost.

Only the keys the JTSM path reads carry defaults (jtsm_amd/config/defaults.py).  Unlike yacs, merging a
file may introduce keys that have no default: the reference's YAMLs set many options of subsystems that
are out of scope here (solver, datasets, test-time augmentation), and they must still load unchanged.
"""
import ast
import copy
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else self._decode(v)

    @staticmethod
    def _decode(v):
        """YAML keeps python literals such as `(60000, 80000)` as strings; evaluate them like yacs does."""
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    # attribute access ---------------------------------------------------------------------------
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        if self.__dict__.get("_frozen", False):
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    # yacs-like API -------------------------------------------------------------------------------
    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self.__dict__["_frozen"]

    def _set_frozen(self, flag):
        self.__dict__["_frozen"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__["_frozen"] = self.__dict__["_frozen"]
        return out

    def dump(self, **kwargs):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(plain(self), **kwargs)

    @staticmethod
    def _coerce(new, old):
        """Keep the declared type where a YAML literal is a near miss (tuple vs list, int vs float)."""
        if old is None or new is None or type(new) is type(old):
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        if isinstance(old, (list, tuple)) and isinstance(new, str):
            try:
                return type(old)(yaml.safe_load(new.replace("(", "[").replace(")", "]")))
            except Exception:
                return new
        return new

    def merge_from_other_cfg(self, other):
        if self.is_frozen():
            raise AttributeError("cannot merge into a frozen CfgNode")
        for k, v in other.items():
            if k == BASE_KEY:
                continue
            if isinstance(v, dict):
                v = v if isinstance(v, CfgNode) else CfgNode(v)
                if isinstance(self.get(k), CfgNode):
                    self[k].merge_from_other_cfg(v)
                else:
                    self[k] = v.clone()
            else:
                self[k] = self._coerce(copy.deepcopy(v), self.get(k))

    @classmethod
    def load_yaml_with_base(cls, filename, allow_unsafe=False):
        """Load a YAML file, resolving `_BASE_` (relative to the file, `~` expanded) depth-first so the
        child overrides its base (config.py:29-70)."""
        with open(filename, "r") as f:
            text = f.read()
        # yacs configs write tuples as python literals, e.g. STEPS: (60000, 80000)
        cfg = yaml.safe_load(text) or {}
        if BASE_KEY in cfg:
            base_file = cfg.pop(BASE_KEY)
            if base_file.startswith("~"):
                base_file = os.path.expanduser(base_file)
            if not os.path.isabs(base_file):
                base_file = os.path.join(os.path.dirname(filename), base_file)
            base = cls.load_yaml_with_base(base_file)

            def merge_a_into_b(a, b):
                for k, v in a.items():
                    if isinstance(v, dict) and isinstance(b.get(k), dict):
                        merge_a_into_b(v, b[k])
                    else:
                        b[k] = v
            merge_a_into_b(cfg, base)
            return base
        return cfg

    def merge_from_file(self, cfg_filename, allow_unsafe=True):
        self.merge_from_other_cfg(CfgNode(self.load_yaml_with_base(cfg_filename)))

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v.replace("(", "[").replace(")", "]")) if v[:1] in "([" else yaml.safe_load(v)
                except Exception:
                    pass
            node[parts[-1]] = self._coerce(v, node.get(parts[-1]))


def get_cfg():
    """A fresh copy of the defaults (detectron2/config/config.py:76-86), with the WSL keys added
    (projects/WSL/wsl/config/defaults.py:7-73) since this package exists for that project."""
    from .defaults import _C, add_wsl_config

    cfg = _C.clone()
    add_wsl_config(cfg)
    return cfg
