"""Step-mode mixin and stateful-module base of the neuron surface.

Behavioural contract taken from SJ/activation_based/base.py:52-117 (``StepModule``) and :153-447
(``MemoryModule``), re-implemented for this build:

* ``step_mode`` is ``'s'`` (single step, [N, ...]) or ``'m'`` (multi step, [T, N, ...]); anything else -> ValueError.
* a *memory* (e.g. the membrane potential ``v``) is an attribute registered with a reset value; ``reset()`` puts
  an independent copy of the reset value back (for ``v``: the python float 0.0), ``.to()/.cuda()`` move tensor
  memories with the module, memories are NOT part of ``state_dict``.
* ``backend`` must be one of ``supported_backends`` (the reference's operator-plugin switch, :199-208).  In this
  build every backend name is served by the same engine, ``libspkdiff.so``.
"""
import copy

import torch
import torch.nn as nn

_BACKENDS_SERVED_BY_HIP = ('torch', 'hip')


def check_backend_library(backend: str):
    if backend in _BACKENDS_SERVED_BY_HIP:
        return
    if backend == 'cupy':
        raise ImportError('CuPy is not part of the MI355X build; libspkdiff (HIP) is the native backend.')
    raise NotImplementedError(backend)


class StepModule:
    _STEP_MODES = ('s', 'm')

    def supported_step_mode(self):
        return self._STEP_MODES

    @property
    def step_mode(self):
        return self._step_mode

    @step_mode.setter
    def step_mode(self, value: str):
        if value not in self.supported_step_mode():
            raise ValueError(f'step_mode can only be {self.supported_step_mode()}, but got "{value}"!')
        self._step_mode = value


class _Slot:
    """One memory: current value + the value ``reset()`` restores."""
    __slots__ = ('value', 'reset_value')

    def __init__(self, value):
        self.value = value
        self.reset_value = copy.deepcopy(value)


class MemoryModule(nn.Module, StepModule):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, '_slots', {})
        self._backend = 'torch'
        self.step_mode = 's'

    # ---- backend switch --------------------------------------------------------------------------------------
    @property
    def supported_backends(self):
        return ('torch',)

    @property
    def backend(self):
        return self._backend

    @backend.setter
    def backend(self, value: str):
        if value not in self.supported_backends:
            raise NotImplementedError(f'{value} is not a supported backend of {self._get_name()}!')
        check_backend_library(value)
        self._backend = value

    # ---- dispatch ----------------------------------------------------------------------------------------------
    def single_step_forward(self, x, *args, **kwargs):
        raise NotImplementedError

    def multi_step_forward(self, x_seq, *args, **kwargs):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        mode = self.step_mode
        if mode == 'm':
            return self.multi_step_forward(*args, **kwargs)
        if mode == 's':
            return self.single_step_forward(*args, **kwargs)
        raise ValueError(mode)

    def extra_repr(self):
        return f'step_mode={self.step_mode}, backend={self.backend}'

    # ---- memories -----------------------------------------------------------------------------------------------
    def register_memory(self, name: str, value):
        if name in self._slots or hasattr(self, name):
            raise AssertionError(f'{name} has been set as a member variable!')
        self._slots[name] = _Slot(value)

    def set_reset_value(self, name: str, value):
        self._slots[name].reset_value = copy.deepcopy(value)

    def reset(self):
        for slot in self._slots.values():
            slot.value = copy.deepcopy(slot.reset_value)

    def memories(self):
        return (slot.value for slot in self._slots.values())

    def named_memories(self):
        return ((name, slot.value) for name, slot in self._slots.items())

    def detach(self):
        for slot in self._slots.values():
            if torch.is_tensor(slot.value):
                slot.value.detach_()

    def __getattr__(self, name: str):
        slots = self.__dict__.get('_slots')
        if slots is not None and name in slots:
            return slots[name].value
        return super().__getattr__(name)

    def __setattr__(self, name: str, value):
        slots = self.__dict__.get('_slots')
        if slots is not None and name in slots:
            slots[name].value = value
        else:
            super().__setattr__(name, value)

    def __delattr__(self, name):
        slots = self.__dict__.get('_slots')
        if slots is not None and name in slots:
            del slots[name]
        else:
            super().__delattr__(name)

    def _apply(self, fn, *args, **kwargs):
        # tensor memories follow .to()/.cuda()/.float(); reset values are left untouched
        for slot in self._slots.values():
            if torch.is_tensor(slot.value):
                slot.value = fn(slot.value)
        return super()._apply(fn, *args, **kwargs)
