"""The RCCL entry points of the C ABI (csrc/comm.hip: ppsci_comm_unique_id / _init / ppsci_allreduce_sum / ppsci_allgather /
_destroy).  One GPU is available to the tests, so the communicator has world size 1: the calls must load librccl lazily,
create the communicator, leave a buffer unchanged under SUM, copy it under all-gather, and tear down.  Multi-rank data
parallelism is covered on CPU by tests/test_distributed.py (gloo, world size 2) through the same Engine.allreduce()."""
import ctypes as C

import numpy as np
import pytest
import torch


def test_comm_entry_points_report_unavailable_in_the_emulator():
    from paddlescience_amd import _lib as L
    from tests.emu import build_emu

    build_emu.inject()
    try:
        lib = L.lib()
        assert lib.ppsci_comm_world_size() == 0
        buf = C.create_string_buffer(128)
        assert lib.ppsci_comm_unique_id(buf) != 0
        assert b"emulator" in lib.ppsci_last_error()
        assert lib.ppsci_comm_destroy() == 0
    finally:
        L._inject_for_tests(None)


@pytest.mark.gpu
def test_comm_world_size_one_roundtrip():
    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    L._inject_for_tests(None)
    lib = L.lib()
    assert lib.ppsci_comm_world_size() == 0
    buf = C.create_string_buffer(128)
    L.check(lib.ppsci_comm_unique_id(buf))
    L.check(lib.ppsci_comm_init(0, 1, buf))
    try:
        assert lib.ppsci_comm_world_size() == 1
        assert lib.ppsci_comm_init(0, 1, buf) != 0  # one communicator per process
        x = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
        ref = x.clone()
        L.check(lib.ppsci_allreduce_sum(hp._p(x), x.numel(), hp._stream_ptr(x)))
        y = torch.empty_like(x)
        L.check(lib.ppsci_allgather(hp._p(x), hp._p(y), x.numel(), hp._stream_ptr(x)))
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(y, ref)
    finally:
        L.check(lib.ppsci_comm_destroy())
    assert lib.ppsci_comm_world_size() == 0
