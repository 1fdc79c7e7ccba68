"""Weight / bias injection by class substitution (the mechanism of the reference's qsparse/imitation.py:7-72).

``imitate(module, "prune", op)`` registers ``op`` as ``module.prune`` and swaps the module's class for a
one-off subclass whose ``weight`` and ``bias`` are *properties*: reading them applies the operator
to the value the previous class would have returned.  Stacking works because each new subclass
derives from the previous one, so ``quantize(prune(conv))`` reads ``quantize(prune(raw_weight))``;
the raw tensors stay in ``module._parameters`` where optimizers and ``state_dict`` find them.
"""
from typing import Callable, Optional

import torch.nn as nn


def _identity(value):
    return value


def imitate(human: nn.Module, name: str, thing: Callable, bias_thing: Optional[Callable] = None) -> nn.Module:
    """transform ``human.weight`` with ``thing`` and ``human.bias`` with ``bias_thing`` (identity when
    omitted); the operators become the attributes ``<name>`` and ``<name>_bias`` of the module.

    The returned object is the same module instance with a new class named like the old one.  As in
    the reference it is not picklable with the stock pickler (the class is created at run time)."""
    base = type(human)

    def inherited(self, attr):
        # a property on the previous class (an earlier imitation) wins over the raw parameter
        if hasattr(base, attr):
            return getattr(base, attr).__get__(self)
        return self._parameters[attr]

    setattr(human, name, thing)
    setattr(human, name + "_bias", bias_thing if bias_thing is not None else _identity)

    def read_weight(self):
        return getattr(self, name)(inherited(self, "weight"))

    def read_bias(self):
        return getattr(self, name + "_bias")(inherited(self, "bias"))

    # `_qs_imitation`: which operator this subclass reads through (outermost first along the MRO) -- the multi-tensor weight
    # path (batch.py) needs the order of a layer's operators to know what a quantizer's input is
    patched = type(base.__name__, (base,), {"weight": property(read_weight), "bias": property(read_bias), "_qs_imitation": name})
    human.__class__ = patched
    return human
