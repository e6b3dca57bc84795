"""Instances — per-image container of equally long fields (boxes, scores, labels ...) with the call
surface model code expects from detectron2/structures/instances.py: attribute get/set, has/get/set/remove,
indexing by int / slice / index or mask tensor, len(), to(), cat()."""
from typing import Any, Dict, List, Tuple

import torch


def _length(value):
    return len(value)


def _concat(values):
    head = values[0]
    if isinstance(head, torch.Tensor):
        return torch.cat(values, dim=0)
    if isinstance(head, list):
        return [x for v in values for x in v]
    joiner = getattr(type(head), "cat", None)
    if joiner is None:
        raise ValueError("Unsupported type {} for concatenation".format(type(head)))
    return joiner(values)


class Instances:
    __slots__ = ("_image_size", "_fields")

    def __init__(self, image_size: Tuple[int, int], **fields: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for name, value in fields.items():
            self.set(name, value)

    # ---- field access ---------------------------------------------------------------------------
    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def set(self, name: str, value: Any) -> None:
        if self._fields:
            have, got = len(self), _length(value)
            assert have == got, "Adding a field of length {} to a Instances of length {}".format(got, have)
        self._fields[name] = value

    def get(self, name: str) -> Any:
        return self._fields[name]

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def __setattr__(self, name: str, value: Any) -> None:
        if name.startswith("_"):          # the two slots (copy / pickle restore them this way)
            object.__setattr__(self, name, value)
        else:
            self.set(name, value)

    def __getattr__(self, name: str) -> Any:
        if name.startswith("_"):          # a slot not set yet (object under construction by copy / pickle)
            raise AttributeError(name)
        fields = object.__getattribute__(self, "_fields")
        if name in fields:
            return fields[name]
        raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))

    # ---- whole-container operations -------------------------------------------------------------
    def _rebuild(self, fn) -> "Instances":
        out = Instances(self._image_size)
        for name, value in self._fields.items():
            out.set(name, fn(value))
        return out

    def to(self, *args: Any, **kwargs: Any) -> "Instances":
        return self._rebuild(lambda v: v.to(*args, **kwargs) if hasattr(v, "to") else v)

    def __getitem__(self, item) -> "Instances":
        if isinstance(item, int):
            n = len(self)
            if not -n <= item < n:
                raise IndexError("Instances index out of range!")
            item = slice(item, None, n)   # keeps the result 1 element long
        return self._rebuild(lambda v: v[item])

    def __len__(self) -> int:
        for value in self._fields.values():
            return _length(value)
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0 and all(isinstance(i, Instances) for i in instance_lists)
        first = instance_lists[0]
        if len(instance_lists) == 1:
            return first
        out = Instances(first.image_size)
        for name in first._fields:
            out.set(name, _concat([inst.get(name) for inst in instance_lists]))
        return out

    def __repr__(self) -> str:
        body = ", ".join("{}: {}".format(k, v) for k, v in self._fields.items())
        return "Instances(num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            len(self) if self._fields else 0, self._image_size[0], self._image_size[1], body)

    __str__ = __repr__
