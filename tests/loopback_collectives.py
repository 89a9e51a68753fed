"""TEST INFRASTRUCTURE: a device-side collective transport for several shards living in ONE process on ONE GPU (one host
thread and one HIP stream per shard).  Every box this repository has seen has a single MI355X, RCCL refuses two ranks on one
device, and the gloo hooks synchronise the stream around every collective -- so without this the engine's pipeline (the host
queueing two population updates ahead, guarded no-op steps behind a fired resample test, collectives on device pointers in
stream order) would never run with device collectives in flight for world > 1.

The hooks get raw device pointers and the library's stream (device_buffers = 1, like RCCL / torch "nccl") and never wait
for the stream on the host: each collective copies the shard's contribution into a double-buffered staging slot, records
an event, meets the other shards at a host barrier (so that every event is recorded before anyone waits on it), makes its
stream wait for the peers' events and combines the slots IN RANK ORDER on its own stream."""
import ctypes as C
import threading

import torch


class Loopback:
    def __init__(self, world, device=0, max_doubles=1 << 22):
        self.world, self.device = world, device
        self.barrier = threading.Barrier(world)
        dev = f"cuda:{device}"
        self.stage = [[torch.zeros(max_doubles, dtype=torch.float64, device=dev) for _ in range(2)] for _ in range(world)]
        self.staged = [[torch.cuda.Event() for _ in range(2)] for _ in range(world)]
        self.read_done = [[None, None] for _ in range(world)]
        self.calls = [0] * world
        self.meta = [[None, None] for _ in range(world)]          # per rank and slot: send counts of an alltoallv
        self.max_doubles = max_doubles
        self.host_waits = 0

    @staticmethod
    def _tensor(ptr, n, device):
        class P:
            pass
        p = P()
        p.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}
        return torch.as_tensor(p, device=f"cuda:{device}")

    def hooks(self, rank):
        W, dev = self.world, self.device

        def begin(stream, send_ptr, n_send, meta=None):
            k = self.calls[rank]
            self.calls[rank] += 1
            slot = k % 2
            s = torch.cuda.ExternalStream(int(stream), device=f"cuda:{dev}")
            assert n_send <= self.max_doubles
            with torch.cuda.stream(s):
                for p in range(W):                       # the peers must have finished reading this slot (collective k - 2)
                    ev = self.read_done[p][slot]
                    if p != rank and ev is not None:
                        ev.wait(s)
                if n_send:
                    self.stage[rank][slot][:n_send].copy_(self._tensor(send_ptr, n_send, dev), non_blocking=True)
                self.meta[rank][slot] = meta
                self.staged[rank][slot].record(s)
            self.barrier.wait()                          # every shard has RECORDED (not executed) its staging copy
            with torch.cuda.stream(s):
                for p in range(W):
                    if p != rank:
                        self.staged[p][slot].wait(s)
            return s, slot

        def end(s, slot):
            ev = torch.cuda.Event()
            ev.record(s)
            self.read_done[rank][slot] = ev
            self.barrier.wait()                          # nobody re-records a slot event before everyone has queued its waits

        def allreduce(ctx, buf, count, stream):
            try:
                s, slot = begin(stream, buf, count)
                with torch.cuda.stream(s):
                    out = self._tensor(buf, count, dev)
                    acc = self.stage[0][slot][:count].clone()
                    for p in range(1, W):                # rank order: every shard gets bitwise the same sum
                        acc += self.stage[p][slot][:count]
                    out.copy_(acc, non_blocking=True)
                end(s, slot)
                return 0
            except Exception as e:                       # never raise through the C frame
                print(f"[loopback] allreduce failed: {e!r}", flush=True)
                return -1

        def allgather(ctx, send, recv, count, stream):
            try:
                s, slot = begin(stream, send, count)
                with torch.cuda.stream(s):
                    out = self._tensor(recv, count * W, dev)
                    for p in range(W):
                        out[p * count:(p + 1) * count].copy_(self.stage[p][slot][:count], non_blocking=True)
                end(s, slot)
                return 0
            except Exception as e:
                print(f"[loopback] allgather failed: {e!r}", flush=True)
                return -1

        def alltoallv(ctx, send, send_counts, recv, recv_counts, nranks, stream):
            try:
                sc = [int(send_counts[p]) for p in range(nranks)]
                rc = [int(recv_counts[p]) for p in range(nranks)]
                s, slot = begin(stream, send, sum(sc), meta=sc)
                with torch.cuda.stream(s):
                    out = self._tensor(recv, max(sum(rc), 1), dev)
                    o = 0
                    for p in range(W):
                        psc = self.meta[p][slot]
                        off = sum(psc[:rank])
                        assert psc[rank] == rc[p], (psc, rc)
                        if rc[p]:
                            out[o:o + rc[p]].copy_(self.stage[p][slot][off:off + rc[p]], non_blocking=True)
                        o += rc[p]
                end(s, slot)
                return 0
            except Exception as e:
                print(f"[loopback] alltoallv failed: {e!r}", flush=True)
                return -1

        return allreduce, allgather, alltoallv
