"""Collective hooks for a sharded population: one process per GPU under torch.distributed.
Backend "nccl" IS RCCL on ROCm: the hooks get device pointers and run on the library's HIP
stream (no host round trip).  Backend "gloo" gets host pointers (the library stages them);
it exists for tests on one GPU / no GPU.

Per population update the engine issues ONE allreduce of the fused sums
(1 + 2s + d + d(d+1)/2 doubles: n_accept, Σu, Σρ, Σθ, Σθθᵀ); an allgather of the population
only when a resample triggers (SimulatedAnnealingABC.jl:340) and, for DifferentialEvolution /
StretchMove, an allgather of θ per half batch so partners are drawn from all shards."""
from __future__ import annotations

import ctypes as C

import numpy as np


class _DevPtr:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def _host_tensor(ptr, n):
    import torch
    arr = np.ctypeslib.as_array((C.c_double * int(n)).from_address(int(ptr)))
    return torch.from_numpy(arr)


def make_hooks(device, group=None):
    """Return (allreduce, allgather, alltoallv, device_buffers) for sabc_set_collectives / sabc_set_alltoallv."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    world = dist.get_world_size(group)
    on_device = backend == "nccl"

    def tensor(ptr, n):
        if on_device:
            return torch.as_tensor(_DevPtr(ptr, n), device=f"cuda:{device}")
        return _host_tensor(ptr, n)

    def stream_ctx(stream):
        if on_device and stream:
            return torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=f"cuda:{device}"))
        import contextlib
        return contextlib.nullcontext()

    def allreduce(ctx, buf, count, stream):
        try:
            with stream_ctx(stream):
                dist.all_reduce(tensor(buf, count), group=group)
            return 0
        except Exception as e:   # never raise through the C frame
            print(f"[sabc] allreduce hook failed: {e!r}", flush=True)
            return -1

    def allgather(ctx, send, recv, count, stream):
        try:
            with stream_ctx(stream):
                out = tensor(recv, count * world)
                dist.all_gather(list(out.chunk(world)), tensor(send, count), group=group)
            return 0
        except Exception as e:
            print(f"[sabc] allgather hook failed: {e!r}", flush=True)
            return -1

    def alltoallv(ctx, send, send_counts, recv, recv_counts, nranks, stream):
        try:
            sc = [int(send_counts[p]) for p in range(nranks)]
            rc = [int(recv_counts[p]) for p in range(nranks)]
            with stream_ctx(stream):
                inp = tensor(send, max(sum(sc), 1))[: sum(sc)]
                out = tensor(recv, max(sum(rc), 1))[: sum(rc)]
                dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc, group=group)
            return 0
        except Exception as e:
            print(f"[sabc] alltoallv hook failed: {e!r}", flush=True)
            return -1

    return allreduce, allgather, alltoallv, on_device


def _all_ok(ok, device, group=None):
    """True on every rank iff `ok` is true on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.cuda(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


def try_p2p(handle, device, group=None, timeout_ms=None):
    """Switch `handle` to the peer-to-peer transport (the shards of one node exchange through each other's HBM: one launch
    per population update instead of reduce -> allreduce -> control; partners and resampled rows read from their owners) if
    every rank can: world <= 8, every peer's memory maps (hipIpc) and the library's self-test -- known rows through the slots,
    and patterns written into the populations read back through the mappings -- passes everywhere.  The agreement between
    the ranks happens INSIDE the library (sabc_comm_p2p_setup, over the collectives already installed, which stay as the
    fallback): a Julia or plain-C host gets the same guarantee.  Returns True when the handle now runs peer to peer (the same
    answer on every rank)."""
    import os
    if not timeout_ms:
        timeout_ms = float(os.environ.get("SABC_P2P_TIMEOUT_MS", "0") or 0)      # bound of every peer-to-peer wait (default 5000)
    if timeout_ms:
        handle.p2p_set_timeout(timeout_ms)
    ok = handle.p2p_setup()
    if not ok and handle.p2p_setup_note and os.environ.get("SABC_P2P_VERBOSE"):
        print(f"[sabc] {handle.p2p_setup_note}", flush=True)
    return ok


def install_collectives(handle, device, group=None, prefer=None, alltoallv=True, p2p=None):
    """Give `handle` its allreduce / allgather.

    prefer="rccl": RCCL bound inside the library (ncclAllReduce / ncclAllGather enqueued on the
    library's stream with no Python in the per-update path); the unique id travels over the
    existing torch.distributed group.  Used by default when the group's backend is "nccl"; if any
    rank fails to set it up or the self-test fails, every rank falls back to the hooks.
    prefer="hooks": torch.distributed collectives through the C-ABI hooks (the only choice for "gloo").
    alltoallv=False leaves the personalised exchange out (hooks only): the resample then allgathers the whole population.
    p2p (default: env SABC_P2P, else on for "nccl" groups): after the collectives are in place, try the peer-to-peer
    transport on top of them (try_p2p); they stay installed as the fallback a failed peer-to-peer call returns to.
    Returns the transport in use: "p2p", "rccl", "hooks-nccl" (device pointers, Python in the per-update path) or
    "hooks-gloo" (host staged; for tests); `handle.fallback_transport` names what sits underneath "p2p"."""
    base = _install_base(handle, device, group, prefer, alltoallv)
    handle.fallback_transport = base
    import os
    import torch.distributed as dist
    if p2p is None:
        p2p = os.environ.get("SABC_P2P", "1" if dist.get_backend(group) == "nccl" else "0") not in ("0", "", "off")
    if p2p and try_p2p(handle, device, group) and handle.p2p_active:    # (a host-callback simulator keeps to the collectives)
        return "p2p"
    return base


def _install_base(handle, device, group=None, prefer=None, alltoallv=True):
    import os
    import torch.distributed as dist
    from .handle import rccl_unique_id

    if prefer is None:
        prefer = os.environ.get("SABC_COLLECTIVES", "rccl" if dist.get_backend(group) == "nccl" else "hooks")
    if prefer == "rccl":
        ok = True
        try:
            box = [rccl_unique_id() if dist.get_rank(group) == 0 else None]
        except Exception as e:
            print(f"[sabc] RCCL unique id failed: {e!r}", flush=True)
            box, ok = [None], False
        dist.broadcast_object_list(box, src=0, group=group)
        ok = ok and box[0] is not None
        if _all_ok(ok, device, group):
            try:
                handle.comm_init_rccl(box[0])
                handle.comm_selftest()
            except Exception as e:
                print(f"[sabc] RCCL setup failed on rank {dist.get_rank(group)}: {e!r}", flush=True)
                ok = False
            if _all_ok(ok, device, group):
                return "rccl"
    ar, ag, a2a, on_device = make_hooks(device, group)
    handle.set_collectives(ar, ag, on_device)
    if alltoallv:
        handle.set_alltoallv(a2a)
    handle.comm_selftest()
    return "hooks-nccl" if on_device else "hooks-gloo"
