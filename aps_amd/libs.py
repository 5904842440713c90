"""
Name -> class registries and the aps_* lookup functions of the reference plugin surface
(aps/libs.py:17-186): same registry names ("asr", "sse", "task", "loader", "trainer",
"transform"), same `@ApsRegisters.X.register("alias")` decorator, same `name@impl` strings and the
`path.py:Class` dynamic import.  Only the sub-modules this build implements are imported by
`import_all`; asking for anything else raises RuntimeError exactly like an unknown name does in
the reference.
"""
import importlib
import warnings
from importlib.machinery import SourceFileLoader
from os.path import basename
from typing import Any, List

import torch.nn as nn


class Register(dict):
    """dict with a decorator-style `register` (duplicate registration only warns, so a plugin may
    override an alias by registering after import, as in the reference)."""

    def __init__(self, name: str) -> None:
        super().__init__()
        self.name = name

    def register(self, alias: str):

        def add(obj):
            if alias in self:
                warnings.warn(f"{alias}: {obj} has already been registered in {self.name}")
            self[alias] = obj
            return obj

        return add


class Module(object):

    def __init__(self, base: str, module: List[str]) -> None:
        self.base = base
        self.module = module

    def import_all(self):
        for sub in self.module:
            importlib.import_module(".".join([self.base, sub]))


class ApsRegisters(object):
    asr = Register("asr")
    sse = Register("sse")
    task = Register("task")
    loader = Register("loader")
    trainer = Register("trainer")
    transform = Register("transform")
    container = [asr, sse, task, loader, trainer, transform]


class ApsModules(object):
    # sub-modules that exist in this build (the hot path of SURVEY.md section 8)
    asr = Module("aps_amd.asr", ["filter.mvdr", "ctc", "att", "enh_att"])
    sse = Module("aps_amd.sse", ["bss.dccrn"])
    task = Module("aps_amd.task", ["sse", "ml"])
    transform = Module("aps_amd.transform", ["asr", "enh"])


def dynamic_importlib(sstr: str) -> Any:
    """`toy_nnet.py:ToyNet` -> class object"""
    path, cls_name = sstr.split(":")
    pkg_name = basename(path).split(".")[0]
    libs = SourceFileLoader(pkg_name, path).load_module(pkg_name)
    if hasattr(libs, cls_name):
        return getattr(libs, cls_name)
    raise ImportError(f"Import {sstr} failed")


def aps_specific_nnet(nnet: str, nnet_cls: Register) -> Any:
    if nnet in nnet_cls:
        return nnet_cls[nnet]
    if ":" in nnet:
        return dynamic_importlib(nnet)
    raise RuntimeError(f"Unsupported nnet: {nnet}")


def aps_transform(name: str) -> Any:
    ApsModules.transform.import_all()
    return aps_specific_nnet(name, ApsRegisters.transform)


def aps_asr_nnet(nnet: str) -> Any:
    ApsModules.asr.import_all()
    return aps_specific_nnet(nnet, ApsRegisters.asr)


def aps_sse_nnet(nnet: str) -> Any:
    ApsModules.sse.import_all()
    return aps_specific_nnet(nnet, ApsRegisters.sse)


def aps_nnet(nnet: str) -> Any:
    nnet_cls, _ = nnet.split("@")
    if nnet_cls in ["rt_sse", "sse"]:
        return aps_sse_nnet(nnet)
    if nnet_cls in ["streaming_asr", "asr"]:
        return aps_asr_nnet(nnet)
    raise RuntimeError(f"Unknown type of the network: {nnet_cls}")


def aps_task(task: str, nnet: nn.Module, **kwargs) -> nn.Module:
    ApsModules.task.import_all()
    if task in ApsRegisters.task:
        impl = ApsRegisters.task[task]
    elif ":" in task:
        impl = dynamic_importlib(task)
    else:
        raise RuntimeError(f"Unsupported task: {task}")
    return impl(nnet, **kwargs)
