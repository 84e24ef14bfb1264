"""Model registry: mirror of ``Factory`` / ``ModelFactory`` (src/mon/core/factory.py:26-438).

Same surface and name resolution as the reference: a nested dict ``arch -> name -> class``
(factory.py:300-328) and ``build(name=..., config=..., **kwargs)`` that tries the raw, kebab,
snake and pascal spellings of ``name`` across all archs (factory.py:363-372).  The reference
uses the ``humps`` package for the case conversions; it is not available here, so the three
conversions it calls are restated below.
"""
from __future__ import annotations

import copy
import inspect
import re
from typing import Any

_SPLIT_RE = re.compile(r"([a-z0-9])([A-Z])")
_ACRONYM_RE = re.compile(r"([A-Z]+)([A-Z][a-z0-9])")


def _is_pascal_or_camel(s: str) -> bool:
    return bool(re.search(r"[A-Z]", s)) and "_" not in s and "-" not in s


def depascalize(s: str) -> str:
    """humps.depascalize: PascalCase/camelCase -> snake_case; other strings pass through."""
    if not isinstance(s, str) or not s:
        return s
    if s.isupper() or s.isnumeric():
        return s
    t = _ACRONYM_RE.sub(r"\1_\2", s)
    t = _SPLIT_RE.sub(r"\1_\2", t)
    return t.replace("-", "_").lower()


def pascalize(s: str) -> str:
    """humps.pascalize: snake_case / kebab-case -> PascalCase (non-word chars kept)."""
    if not isinstance(s, str) or not s:
        return s
    if s.isupper() or s.isnumeric():
        return s
    t = re.sub(r"(?:^|[_\-\s]+)([^_\-\s])", lambda m: m.group(1).upper(), s)
    return t[0].upper() + t[1:] if t else t


def kebabize(s: str) -> str:
    """humps.kebabize: -> kebab-case."""
    if not isinstance(s, str) or not s:
        return s
    return depascalize(s).replace("_", "-")


class RegistryLookupError(KeyError, ValueError):
    """Unknown arch/name.  The reference raises ``KeyError`` from ``self[arch][name]``
    (factory.py:380; its ``ValueError`` guard at :374-378 never fires); both are honoured."""

    def __str__(self):
        return self.args[0] if self.args else ""


class Factory(dict):
    """factory.py:26-230 -- flat ``name -> class`` registry."""

    def __init__(self, name: str, mapping: dict = None, *args, **kwargs):
        if name in [None, "None", ""]:
            raise ValueError("`name` must be given to create a valid factory object.")
        self.name = name
        super().__init__(mapping or {})

    def __repr__(self) -> str:
        return self.__class__.__name__ + f"(name={self.name}, items={dict(self)})"

    def register(self, name: str = None, module: Any = None, replace: bool = False):
        if not (name is None or isinstance(name, str)):
            raise TypeError(f"`name` must be a `str`, but got {type(name)}.")
        if module:
            self.register_module(module_cls=module, module_name=name, replace=replace)
            return module

        def _register(cls):
            self.register_module(module_cls=cls, module_name=name, replace=replace)
            return cls
        return _register

    def register_module(self, module_cls: Any, module_name: str = None, replace: bool = False):
        if not inspect.isclass(module_cls):
            raise ValueError(f"`module_cls` must be a class interface, but got {type(module_cls)}.")
        module_name = module_name or kebabize(module_cls.__name__)
        if replace or module_name not in self:
            self[module_name] = module_cls

    def build(self, name: str = None, config: dict = None, to_dict: bool = False, **kwargs):
        if (name is None and config is None) or (name is None and config and "name" not in config):
            return None
        if config:
            config_ = copy.deepcopy(config)
            name = name or config_.pop("name", None)
            kwargs |= config_
        if name is None or name not in self:
            raise RegistryLookupError(f"`name` must be a valid keyword inside the registry, but got {name}.")
        obj = self[name](**kwargs)
        if getattr(obj, "name", None) is None:
            obj.name = depascalize(pascalize(name))
        return {f"{name}": obj} if to_dict else obj


class ModelFactory(Factory):
    """factory.py:233-438 -- ``arch -> name -> class`` registry."""

    @property
    def archs(self) -> list[str]:
        return list(self.keys())

    @property
    def models(self) -> list[str]:
        return [sub_k for k, d in self.items() if isinstance(d, dict) for sub_k in d]

    def register(self, name: str = None, arch: str = None, module: Any = None, replace: bool = False):
        if not (name is None or isinstance(name, str)):
            raise TypeError(f"`name` must be a `str`, but got {type(name)}.")
        if module:
            self.register_module(module_cls=module, module_name=name, arch_name=arch, replace=replace)
            return module

        def _register(cls):
            self.register_module(module_cls=cls, module_name=name, arch_name=arch, replace=replace)
            return cls
        return _register

    def register_module(self, module_cls: Any, module_name: str = None, arch_name: str = None,
                        replace: bool = False):
        if not inspect.isclass(module_cls):
            raise ValueError(f"`module_cls` must be a class interface, but got {type(module_cls)}.")
        module_name = module_name or kebabize(module_cls.__name__)
        arch_name = arch_name or kebabize(getattr(module_cls, "arch", None))
        arch_name = arch_name or kebabize(module_cls.__name__)
        if arch_name not in self:
            self[arch_name] = {}
        if replace or module_name not in self[arch_name]:
            self[arch_name][module_name] = module_cls

    def build(self, name: str = None, arch: str = None, config: dict = None, to_dict: bool = False, **kwargs):
        if (name is None and config is None) or (name is None and config and "name" not in config):
            return None
        if config:
            config_ = copy.deepcopy(config)
            name = name or config_.pop("name", None)
            kwargs |= config_
        arch = arch or name
        found = False
        if name:
            for n in [name, kebabize(name), depascalize(pascalize(name)), pascalize(name)]:
                for a, models_dict in self.items():
                    if n in models_dict:
                        name, arch, found = n, a, True
                        break
                if found:
                    break
            if not found:
                # humps' acronym handling is not reproduced exactly; fall back to a separator- and
                # case-insensitive comparison so "ZeroDCE_RE" / "zero-dce-re" still resolve.
                key = re.sub(r"[^0-9a-z+]", "", name.lower())
                for a, models_dict in self.items():
                    for n in models_dict:
                        if re.sub(r"[^0-9a-z+]", "", n.lower()) == key:
                            name, arch, found = n, a, True
                            break
                    if found:
                        break
        if not found or arch not in self or name not in self[arch]:
            raise RegistryLookupError(f"`arch` and `name` must be a valid keyword inside the registry, "
                                      f"but got {arch} and {name}.")
        obj = self[arch][name](**kwargs)
        if getattr(obj, "name", None) is None:
            obj.name = depascalize(pascalize(name))
        return {f"{name}": obj} if to_dict else obj

    def build_instances(self, configs: list[Any], to_dict: bool = False, **kwargs):
        if configs is None:
            return None
        if not isinstance(configs, list):
            raise ValueError(f"`configs` must be a `list`, but got {type(configs)}.")
        objs = {} if to_dict else []
        for config in copy.deepcopy(configs):
            if isinstance(config, str):
                name, arch, config = config, None, {}
            elif isinstance(config, dict):
                name = config.pop("name", None)
                arch = config.pop("arch", None)
            else:
                raise ValueError(f"Item inside `configs` must be a `str` or `dict`, but got {type(config)}.")
            obj = self.build(name=name, arch=arch, to_dict=to_dict, **config)
            if obj:
                if to_dict:
                    objs |= obj
                else:
                    objs.append(obj)
        return objs if len(objs) > 0 else None


MODELS = ModelFactory("Models")  # src/mon/globals.py:1224
