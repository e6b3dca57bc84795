"""Name -> object registry with the call surface the reference gets from fvcore
(detectron2/utils/registry.py:4 re-exports fvcore.common.registry.Registry; fvcore is not part of
the reference tree): ``register()`` as decorator or call, ``get(name)``, ``in``."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, "An object named '%s' was already registered in '%s' registry!" % (
            name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def __repr__(self):
        return "Registry of %s: %s" % (self._name, sorted(self._obj_map))
