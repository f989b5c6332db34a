"""Multi-GPU sharding of one long haystack: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" in the CPU tests).

The AhoCorasick (all-matches) path shards naturally: rank g owns a contiguous range of the
haystack, needs the (max_keyword_len-1) units before it (left halo, received from rank g-1), starts at the root,
and keeps the matches whose LAST unit it owns.  Rank-local order is the reference's order, so the concatenation
of the per-rank buffers by rank is the reference's listener-call order for the whole haystack.  The only data
exchange steps are the tiny halo send/recv and the all-gather of the per-shard match buffers (counts first, then
record buffers padded to the largest count); positions stay shard-local int32 in the gathered buffer and become
global int64 positions by adding base[g] = g * units_per_rank (global_records()).
"""
import numpy as np
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def round_up8(v):
    return (int(v) + 7) // 8 * 8


class ShardBuffer:
    """[ pad | own units ] on one device: `pad` = left-halo room rounded up to 8 units so that both the buffer and
    the owned range start 16-byte aligned (the scan kernel loads 16 bytes per lane)."""

    def __init__(self, n_units, halo, device):
        self.n_units = int(n_units)
        self.halo = int(halo)
        self.pad = round_up8(halo)
        self.buf = torch.zeros(self.pad + self.n_units, dtype=torch.int16, device=device)

    @property
    def own(self):
        return self.buf[self.pad:]

    def halo_view(self):
        return self.buf[self.pad - self.halo:self.pad]

    def tail_view(self):
        return self.buf[self.pad + self.n_units - self.halo:]


def exchange_halo(sb, group=None):
    """rank g sends its last `halo` units to rank g+1 and receives its left halo from rank g-1."""
    rank, world = _world(group)
    if world == 1 or sb.halo == 0:
        return
    ops = []
    recv = None
    # bytes on the wire: RCCL has no 16-bit integer type
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, sb.tail_view().contiguous().view(torch.uint8), rank + 1, group))
    if rank > 0:
        recv = torch.empty(2 * sb.halo, dtype=torch.uint8, device=sb.buf.device)
        ops.append(dist.P2POp(dist.irecv, recv, rank - 1, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if recv is not None:
        sb.halo_view().copy_(recv.view(torch.int16))


def allgather_counts(n_local, device, group=None):
    """(world,) int64 match counts on the host."""
    rank, world = _world(group)
    if world == 1:
        return np.array([n_local], dtype=np.int64)
    cnt = torch.tensor([n_local], dtype=torch.int64, device=device)
    counts = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    return counts.cpu().numpy()


def allgather_records(local, n_local, counts_h, group=None, async_op=False):
    """All-gather of the record buffers, padded to the largest count.  local: (cap, cols) int32, first n_local rows
    valid.  Returns (gathered (world, max_n, cols) int32, work handle or None)."""
    rank, world = _world(group)
    if world == 1:
        return local[:n_local].unsqueeze(0), None
    dev = local.device
    max_n = int(counts_h.max())
    cols = local.shape[1]
    if max_n == 0:
        return torch.empty((world, 0, cols), dtype=torch.int32, device=dev), None
    if local.shape[0] >= max_n:
        send = local[:max_n]
    else:
        send = torch.zeros((max_n, cols), dtype=torch.int32, device=dev)
        send[:n_local] = local[:n_local]
    out = torch.empty((world, max_n, cols), dtype=torch.int32, device=dev)
    work = dist.all_gather_into_tensor(out.view(-1), send.contiguous().view(-1), group=group, async_op=async_op)
    return out, (work if async_op else None)


def allgather_matches(local, n_local, group=None):
    """Counts, then records.  Returns (gathered (world, max_n, cols) int32, counts (world,) int64 on host)."""
    counts_h = allgather_counts(n_local, local.device, group)
    out, _ = allgather_records(local, n_local, counts_h, group)
    return out, counts_h


def global_records(gathered, counts, units_per_rank):
    """Concatenate per-rank records in rank order with global int64 positions (reference order of the whole text)."""
    parts = []
    for g, n in enumerate(counts.tolist()):
        r = gathered[g, :n].to(torch.int64).clone()
        r[:, :2] += g * int(units_per_rank)
        parts.append(r)
    return torch.cat(parts) if parts else torch.empty((0, gathered.shape[-1]), dtype=torch.int64)


class ShardedMatcher:
    """One rank's end of the sharded AhoCorasick match: device-resident shard, halo exchange, native scan
    (acgpu_match_device), all-gather of match buffers.

    scan_fn (tests only) replaces the native scan so the plumbing can run under gloo on CPU:
    scan_fn(buffer_units_np, own_begin, own_end, text_begin) -> (n,cols) int32 records, buffer-relative."""

    def __init__(self, automaton, n_units, with_ids=True, cap=1 << 20, device=None, group=None, scan_fn=None, halo=None,
                 overlap=False):
        self.auto = automaton
        self.group = group
        self.rank, self.world = _world(group)
        self.with_ids = with_ids
        self.cols = 3 if with_ids else 2
        if halo is None:
            halo = max(0, automaton.info()["max_keyword_len"] - 1)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if scan_fn is None else torch.device("cpu")
        self.sb = ShardBuffer(n_units, halo, device)
        self.cap = int(cap)
        # overlap: the all-gather of step k runs (on RCCL's stream) while step k+1 scans into the other record buffer
        self.overlap = bool(overlap) and self.world > 1
        # single GPU: the same flag pipelines the calls themselves -- step k+1 is enqueued (acgpu_match_device_begin)
        # before the count of step k is read back, so the GPU never waits for the host between steps
        self.pipeline = bool(overlap) and self.world == 1 and scan_fn is None
        self._ticket = None
        self.outs = [torch.empty((self.cap, self.cols), dtype=torch.int32, device=device)
                     for _ in range(2 if (self.overlap or self.pipeline) else 1)]
        self.out = self.outs[0]
        self._k = 0
        self._pending = None  # (work, gathered, counts) of the all-gather still in flight
        self.scan_fn = scan_fn
        self.last_kernel = ""
        self.gathered = None
        self.counts = None

    def own_ptr(self):
        return self.sb.own.data_ptr()

    def own_units_host(self, k=None):
        v = self.sb.own if k is None else self.sb.own[:k]
        return v.cpu().numpy().view(np.uint16)

    def _scan(self, profile):
        sb = self.sb
        first = self.rank == 0
        last = self.rank == self.world - 1
        if self.scan_fn is not None:
            if first:
                recs = self.scan_fn(sb.own.numpy().view(np.uint16), 0, sb.n_units, True)
            else:
                recs = self.scan_fn(sb.buf.numpy().view(np.uint16), sb.pad, sb.pad + sb.n_units, False)
                recs = recs.copy()
                recs[:, :2] -= sb.pad
            n = len(recs)
            if n > self.out.shape[0]:
                self.cap = max(self.cap, n)
                self.out = self.outs[self._slot] = torch.empty((self.cap, self.cols), dtype=torch.int32)
            self.out[:n] = torch.from_numpy(np.ascontiguousarray(recs[:, :self.cols], dtype=np.int32))
            return n, None
        from . import _native as N
        stream = torch.cuda.current_stream().cuda_stream
        while True:
            if first:  # the pad in front of rank 0's text is not part of the haystack
                n, rc, prof, _ = self.auto.match_device(sb.own.data_ptr(), sb.n_units, self.with_ids, self.out.data_ptr(),
                                                        self.out.shape[0], own=(0, sb.n_units), text_begin=True, text_end=last,
                                                        stream=stream, profile=profile)
            else:
                n, rc, prof, _ = self.auto.match_device(sb.buf.data_ptr(), sb.pad + sb.n_units, self.with_ids,
                                                        self.out.data_ptr(), self.out.shape[0], own=(sb.pad, sb.pad + sb.n_units),
                                                        text_begin=False, text_end=last, stream=stream, profile=profile)
            if rc == N.E_OVERFLOW:
                self.cap = max(self.cap, int(n * 1.25) + 16)
                self.out = self.outs[self._slot] = torch.empty((self.cap, self.cols), dtype=torch.int32,
                                                               device=self.out.device)
                continue
            N.check(rc, "acgpu_match_device")
            break
        if not first and n:
            self.out[:n, :2] -= sb.pad  # buffer-relative -> shard-relative
        if prof:
            self.last_kernel = prof["scan_kernel"]
        return n, prof

    def step(self, profile=False):
        """halo exchange -> scan -> all-gather.  Returns a dict with n_local, n_total and (profile) kernel timings.
        With overlap=True the record all-gather is left in flight; `gathered`/`counts` then describe the last
        COMPLETED step until finish() is called."""
        if self.pipeline:
            return self._step_pipelined(profile)
        exchange_halo(self.sb, self.group)
        self._slot = self._k % len(self.outs)
        self._k += 1
        self.out = self.outs[self._slot]
        n, prof = self._scan(profile)
        counts = allgather_counts(n, self.out.device, self.group)
        if self.overlap:
            self._complete_pending()  # the other buffer's gather: makes this stream wait for it, not the host
            gathered, work = allgather_records(self.out, n, counts, self.group, async_op=True)
            self._pending = (work, gathered, counts)
        else:
            self.gathered, _ = allgather_records(self.out, n, counts, self.group)
            self.counts = counts
        r = {"n_local": int(n), "n_total": int(counts.sum()), "scan_ms": 0.0, "finalize_ms": 0.0}
        if prof:
            r.update(scan_ms=prof["scan_ms"], finalize_ms=prof["finalize_ms"])
        return r

    def _step_pipelined(self, profile):
        """world == 1: enqueue this step, then collect the PREVIOUS one.  Returns the previous step's result dict
        (None for the first call); finish() returns the last one."""
        from . import _native as N
        sb = self.sb
        slot = self._k % 2
        self._k += 1
        out = self.outs[slot]
        tk, rc = self.auto.match_device_begin(sb.own.data_ptr(), sb.n_units, self.with_ids, out.data_ptr(), out.shape[0],
                                              stream=torch.cuda.current_stream().cuda_stream, profile=profile)
        N.check(rc, "acgpu_match_device_begin")
        prev = self._collect(profile)
        self._ticket = (tk, slot, profile)
        return prev

    def _collect(self, profile):
        from . import _native as N
        if self._ticket is None:
            return None
        tk, slot, prof_on = self._ticket
        self._ticket = None
        n, rc, prof = self.auto.match_device_end(tk, profile=prof_on)
        if rc == N.E_OVERFLOW:  # rare: grow both buffers and redo that step synchronously
            self.cap = max(self.cap, int(n * 1.25) + 16)
            self.outs = [torch.empty((self.cap, self.cols), dtype=torch.int32, device=self.outs[0].device) for _ in self.outs]
            self.out = self.outs[slot]
            self._slot = slot
            n, prof = self._scan(prof_on)
        else:
            N.check(rc, "acgpu_match_device_end")
        self.out = self.outs[slot]
        self.gathered, self.counts = self.out[:n].unsqueeze(0), np.array([n], dtype=np.int64)
        if prof:
            self.last_kernel = prof["scan_kernel"]
        r = {"n_local": int(n), "n_total": int(n), "scan_ms": 0.0, "finalize_ms": 0.0}
        if prof:
            r.update(scan_ms=prof["scan_ms"], finalize_ms=prof["finalize_ms"])
        return r

    def _complete_pending(self):
        if self._pending is not None:
            work, gathered, counts = self._pending
            if work is not None:
                work.wait()
            self.gathered, self.counts = gathered, counts
            self._pending = None

    def finish(self):
        """Completes what step() left in flight: the overlapped all-gather (world > 1) or the last pipelined call
        (world == 1; its result dict is returned)."""
        if self.pipeline:
            return self._collect(True)
        self._complete_pending()
        return None

    def global_records(self):
        return global_records(self.gathered, self.counts, self.sb.n_units)
