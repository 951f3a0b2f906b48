"""The collective entry points of the C ABI (csrc/comm.hip: ppsci_comm_unique_id / _init / ppsci_allreduce_sum / ppsci_allgather /
_destroy -- the fused gradient all-reduce of /root/reference/ppsci/solver/train.py:168-171 for hosts without torch.distributed).
On this pool only one GPU is visible, so the hardware test runs a communicator of ONE rank: it proves that librccl is found,
the communicator comes up on the launch stream's device and both collectives run in place (for one rank: the identity); the
multi-rank semantics are RCCL's own.  The argument checks run on any host."""
import ctypes as C

import numpy as np
import pytest
import torch


def _real_lib():
    import __graft_entry__ as g
    from paddlescience_amd import _lib

    return _lib._bind(g.LIB)


def test_collectives_refuse_use_before_init():
    lib = _real_lib()
    assert lib.ppsci_comm_world_size() == 0
    assert lib.ppsci_allreduce_sum(None, 4, None) != 0 and b"comm_init has not been called" in lib.ppsci_last_error()
    assert lib.ppsci_comm_init(2, 2, None) != 0 and b"comm_init" in lib.ppsci_last_error()  # rank outside the world, no id
    assert lib.ppsci_comm_unique_id(None) != 0


@pytest.mark.gpu
def test_single_rank_communicator_runs_both_collectives():
    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    L._inject_for_tests(None)
    lib = L.lib()
    torch.cuda.set_device(0)
    ident = C.create_string_buffer(128)
    L.check(lib.ppsci_comm_unique_id(ident))
    assert any(ident.raw)
    L.check(lib.ppsci_comm_init(0, 1, ident))
    try:
        assert lib.ppsci_comm_world_size() == 1
        assert lib.ppsci_comm_init(0, 1, ident) != 0  # one communicator per process
        g = torch.arange(66_819, dtype=torch.float32, device="cuda") * 1e-3  # the NavierStokes 5 x 128 gradient's size
        ref = g.cpu().numpy().copy()
        L.check(lib.ppsci_allreduce_sum(hp._p(g), g.numel(), hp._stream_ptr(g)))
        out = torch.empty_like(g)
        L.check(lib.ppsci_allgather(hp._p(g), hp._p(out), g.numel(), hp._stream_ptr(g)))
        torch.cuda.synchronize()
        assert np.array_equal(g.cpu().numpy(), ref) and np.array_equal(out.cpu().numpy(), ref)
    finally:
        L.check(lib.ppsci_comm_destroy())
    assert lib.ppsci_comm_world_size() == 0
