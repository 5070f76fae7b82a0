"""CPU-side checks of the host mirror's device-memory rules (no GPU, no compute calls): the pointer a DeviceVector hands
out keeps the vector alive for as long as the pointer object lives, and a freed vector raises instead of yielding a stale
address.  (r03: `ctx.to_device(z).ptr` of a temporary was finalised between `.ptr` and the launch it was passed to -- a
dangling device pointer, a GPU memory access fault)"""
import ctypes
import gc

import pytest


class _FakeLib:
    def __init__(self):
        self.freed = []

    def aggmg_dev_free(self, handle, p):
        self.freed.append(p.value)
        return 0


class _FakeCtx:
    def __init__(self):
        self.lib = _FakeLib()
        self.handle = ctypes.c_void_p(1)

    def check(self, status):
        assert status == 0


def _vector(ctx, addr, n=4):
    from agglomerationmultigrid1d_amd import api
    v = api.DeviceVector.__new__(api.DeviceVector)
    v.ctx, v.n, v._p = ctx, n, ctypes.c_void_p(addr)
    return v


def test_pointer_of_a_temporary_keeps_the_vector_alive():
    ctx = _FakeCtx()
    q = _vector(ctx, 0x1000).ptr          # the vector itself is a temporary
    gc.collect()
    assert ctx.lib.freed == [], "the vector was finalised while its pointer was still in use"
    assert q.value == 0x1000 and isinstance(q, ctypes.c_void_p)
    # what a ctypes call does with it: the argument tuple holds the pointer for the duration of the call
    seen = []

    def call(*args):
        gc.collect()
        seen.append((args[0].value, list(ctx.lib.freed)))

    call(_vector(ctx, 0x2000).ptr)
    assert seen == [(0x2000, [])]
    gc.collect()
    assert ctx.lib.freed == [0x2000]       # released (stream-synchronising free) once the call is over
    del q
    gc.collect()
    assert sorted(ctx.lib.freed) == [0x1000, 0x2000]


def test_pointer_of_a_freed_vector_raises():
    from agglomerationmultigrid1d_amd import ArgumentError
    ctx = _FakeCtx()
    v = _vector(ctx, 0x3000)
    assert v.ptr.value == 0x3000
    v.free()
    assert ctx.lib.freed == [0x3000]
    with pytest.raises(ArgumentError):
        v.ptr
    v.free()                                # idempotent
    assert ctx.lib.freed == [0x3000]


def test_ptr_helper_passes_owning_pointers_through():
    from agglomerationmultigrid1d_amd import api
    ctx = _FakeCtx()
    v = _vector(ctx, 0x4000)
    p = api._ptr(v)
    assert p.value == 0x4000 and p._owner is v
    assert api._ptr(None).value is None
    assert api._ptr(0x5000).value == 0x5000
