"""Configuration dictionaries -> live objects (reference: config.py:29-235).

A configuration is ``{'name': ..., 'class': 'SingletLens', 'params': {...}}`` as the reference's GUI stores
it inside ``.rtt`` project files; parameter values are numbers, strings holding Python literals, 3-vectors,
booleans, or nested configurations for parameters that take an object (transforms, shapes, functors).

Conversion is driven by the *value* first (a nested ``{'class', 'params'}`` dictionary always builds that class)
and by the constructor's annotation second, so files written against the reference's constructors load even
where an annotation here is spelled differently.  The introspection helpers keep the reference's names because
its form builder calls them.
"""
import ast
import inspect
import sys
import typing
from typing import List, Tuple

import torch

from . import elements as _elements
from . import geom as _geom
from . import phys as _phys
from . import rays as _rays

_SCALARS = (int, float, str, bool, list, tuple, dict, torch.Tensor)


def _members(annotation):
    """Flatten Optional/Union annotations into their member types."""
    if typing.get_origin(annotation) is typing.Union:
        out = []
        for arg in typing.get_args(annotation):
            out.extend(_members(arg))
        return out
    return [] if annotation is type(None) else [annotation]


def analyze_type(p_type) -> str:
    """'DTYPE' | 'VEC3' | 'BOOL3' | 'BOOL' | 'CLASS' | 'PRIMITIVE' for a constructor annotation (config.py:29-63)."""
    if p_type == torch.dtype:
        return "DTYPE"
    members = _members(p_type)
    if torch.Tensor in members:
        return "BOOL3" if (List[bool] in members or Tuple[bool, ...] in members) else "VEC3"
    if members == [bool]:
        return "BOOL"
    if any(inspect.isclass(m) and m not in _SCALARS for m in members):
        return "CLASS"
    return "PRIMITIVE"


def get_subclasses(cls) -> set:
    """Every direct and indirect subclass (config.py:66-78)."""
    found, todo = set(), [] if cls is None else [cls]
    while todo:
        for child in todo.pop().__subclasses__():
            if child not in found:
                found.add(child)
                todo.append(child)
    return found


def get_constructor_params(cls) -> dict:
    """name -> (annotation or str, default or None) of ``cls.__init__`` (config.py:81-94)."""
    try:
        sig = inspect.signature(cls.__init__)
    except (ValueError, TypeError):
        return {}
    empty = inspect.Parameter.empty
    return {name: (str if p.annotation is empty else p.annotation, None if p.default is empty else p.default)
            for name, p in sig.parameters.items()
            if name != "self" and p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)}


def get_all_actual_classes(p_type) -> list:
    """All classes named by an annotation (config.py:109-126)."""
    return [m for m in _members(p_type) if inspect.isclass(m)]


def get_actual_class(p_type):
    """First class named by an annotation, or None (config.py:97-106)."""
    classes = get_all_actual_classes(p_type)
    return classes[0] if classes else None


_BASES = (_elements.Element, _rays.Bundle, _geom.Shape, _geom.Surface, _geom.RayTransform, _phys.SurfaceFunction)


def _find_class_by_name(name: str):
    """Subclasses of the package's base types first, then any class of that name in a loaded package module."""
    for base in _BASES:
        if base.__name__ == name:
            return base
        for sub in get_subclasses(base):
            if sub.__name__ == name:
                return sub
    for mod_name, mod in list(sys.modules.items()):
        if mod_name.startswith(__package__ + "."):
            obj = getattr(mod, name, None)
            if inspect.isclass(obj):
                return obj
    return None


def instantiate_from_config(config: dict, device=None, dtype=None):
    """Build the object a ``{'name', 'class', 'params'}`` dictionary describes (config.py:156-177); `device` and
    `dtype` are handed to every constructor that takes them and is not told otherwise by the file."""
    cls = _find_class_by_name(config.get("class", ""))
    if cls is None:
        raise ValueError(f"Cannot find class '{config.get('class', '')}'. Check imports.")
    return _build(cls, config.get("params", {}), device, dtype)


def _is_config(val) -> bool:
    return isinstance(val, dict) and "class" in val and "params" in val


def _build(cls, params: dict, device, dtype):
    ctor = get_constructor_params(cls)
    kwargs = {}
    if device is not None and "device" in ctor and "device" not in params:
        kwargs["device"] = device
    if dtype is not None and "dtype" in ctor and "dtype" not in params:
        kwargs["dtype"] = dtype
    for name, (annotation, _default) in ctor.items():
        if name not in params:
            continue
        val, kind = params[name], analyze_type(annotation)
        if _is_config(val):
            sub = _find_class_by_name(val["class"]) or get_actual_class(annotation)
            if sub is None:
                raise ValueError(f"Cannot find class '{val['class']}' for parameter '{name}' of {cls.__name__}")
            kwargs[name] = _build(sub, val["params"], device, dtype)
        elif kind == "CLASS":
            target = get_actual_class(annotation)
            kwargs[name] = _build(target, val, device, dtype) if isinstance(val, dict) and target else None
        elif kind == "BOOL":
            kwargs[name] = bool(val)
        elif kind == "DTYPE":
            kwargs[name] = torch.float64 if val == "float64" else torch.float32
        elif kind in ("VEC3", "BOOL3"):
            kwargs[name] = None if val is None else list(val)
        else:
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            kwargs[name] = val
    return cls(**kwargs)
