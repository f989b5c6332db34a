"""Multi-GPU sharding of one long haystack: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" in the CPU tests).  SURVEY.md 8e, one row per matcher family:

* AhoCorasick (all matches) shards naturally: rank g owns a contiguous range of the haystack, needs the
  (max_keyword_len-1) units before it (left halo, received from rank g-1), starts at the root, and keeps the matches
  whose LAST unit it owns.
* WholeWord: a word belongs to the rank that owns its FIRST unit; left context 1 unit (is the previous unit a word
  character?), right halo max_keyword_len+1 units from rank g+1 (enough to see that a run is longer than any keyword).
* Longest: the lengths L[pos] need a right halo of max_keyword_len-1 units and are independent per shard; the greedy
  chain pos -> pos + max(L[pos],1) needs each shard's entry position = the previous shard's exit.  Every rank first
  runs its chain speculatively from its own first unit (all ranks in parallel); then ONE int64 travels down the ranks
  (send/recv, world-1 hops) and a rank whose true entry differs re-runs a short window until the true chain leaves the
  window where the speculative one did -- from there on both are the same chain.
* Shortest: a match belongs to the rank that owns its LAST unit (left halo as AhoCorasick); which occurrences are
  reported depends on where matching last restarted (the end of the previous reported match), handed down the ranks
  like the Longest chain position, with the same speculation (no restriction) and window repair.

In all three, rank-local order is the reference's order, so the concatenation of the per-rank buffers by rank is the
reference's listener-call order for the whole haystack.  The data exchange steps are the tiny halo send/recv, the
Longest chain hop, and the all-gather of the per-shard match buffers (counts first, then record buffers padded to the
largest count); positions stay shard-local int32 in the gathered buffer and become global int64 positions by adding
base[g] = g * units_per_rank (global_records()).
"""
import numpy as np
import torch
import torch.distributed as dist

from ._native import MODE_ALL, MODE_LONGEST, MODE_SHORTEST, MODE_WHOLEWORD


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def round_up8(v):
    return (int(v) + 7) // 8 * 8


class ShardBuffer:
    """[ pad | own units | right halo ] on one device: `pad` = left-halo room rounded up to 8 units so that both the
    buffer and the owned range start 16-byte aligned (the scan kernels load 16 bytes per lane)."""

    def __init__(self, n_units, halo, device, right_halo=0):
        self.n_units = int(n_units)
        self.halo = int(halo)
        self.right = int(right_halo)
        self.pad = round_up8(halo)
        self.buf = torch.zeros(self.pad + self.n_units + self.right, dtype=torch.int16, device=device)

    @property
    def own(self):
        return self.buf[self.pad:self.pad + self.n_units]

    def halo_view(self):
        return self.buf[self.pad - self.halo:self.pad]

    def tail_view(self):
        return self.buf[self.pad + self.n_units - self.halo:self.pad + self.n_units]

    def head_view(self):
        return self.buf[self.pad:self.pad + self.right]

    def right_view(self):
        return self.buf[self.pad + self.n_units:]


def exchange_halo(sb, group=None):
    """rank g sends its last `halo` units to rank g+1 / its first `right` units to rank g-1 and receives its left halo
    from rank g-1 / its right halo from rank g+1."""
    rank, world = _world(group)
    if world == 1 or (sb.halo == 0 and sb.right == 0):
        return
    if sb.n_units < max(sb.halo, sb.right):
        raise ValueError("shard of %d units is shorter than its halo (%d, %d)" % (sb.n_units, sb.halo, sb.right))
    dev = sb.buf.device
    ops = []
    recv_l = recv_r = None
    # bytes on the wire: RCCL has no 16-bit integer type
    if sb.halo:
        if rank + 1 < world:
            ops.append(dist.P2POp(dist.isend, sb.tail_view().contiguous().view(torch.uint8), rank + 1, group))
        if rank > 0:
            recv_l = torch.empty(2 * sb.halo, dtype=torch.uint8, device=dev)
            ops.append(dist.P2POp(dist.irecv, recv_l, rank - 1, group))
    if sb.right:
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, sb.head_view().contiguous().view(torch.uint8), rank - 1, group))
        if rank + 1 < world:
            recv_r = torch.empty(2 * sb.right, dtype=torch.uint8, device=dev)
            ops.append(dist.P2POp(dist.irecv, recv_r, rank + 1, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if recv_l is not None:
        sb.halo_view().copy_(recv_l.view(torch.int16))
    if recv_r is not None:
        sb.right_view().copy_(recv_r.view(torch.int16))


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
    """One rank's end of the sharded match of one long haystack: device-resident shard, halo exchange, native scan
    (acgpu_match_device), [Longest: chain hop], all-gather of match buffers.

    scan_fn (tests only) replaces the native scan so the plumbing can run under gloo on CPU; it has the contract of
    acgpu_match_device on one shard:
    scan_fn(view_units_np, own_begin, own_end, text_begin, text_end, chain_entry) -> ((n,cols) int32 records relative
    to the view, chain_exit)."""

    def __init__(self, automaton, n_units, with_ids=True, cap=1 << 20, device=None, group=None, scan_fn=None, halo=None,
                 overlap=False, mode=None, right_halo=None):
        self.auto = automaton
        self.group = group
        self.rank, self.world = _world(group)
        self.with_ids = with_ids
        self.cols = 3 if with_ids else 2
        self.mode = automaton.mode if automaton is not None else (MODE_ALL if mode is None else mode)
        if halo is None or (right_halo is None and self.mode != MODE_ALL):
            max_len = automaton.info()["max_keyword_len"]
            if self.mode in (MODE_ALL, MODE_SHORTEST):
                halo, right_halo = max(0, max_len - 1), 0
            elif self.mode == MODE_WHOLEWORD:
                halo, right_halo = 1, max_len + 1
            else:
                halo, right_halo = 0, max(0, max_len - 1)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if scan_fn is None else torch.device("cpu")
        self.sb = ShardBuffer(n_units, halo, device, right_halo or 0)
        self.cap = int(cap)
        # overlap: the all-gather of step k runs (on RCCL's stream) while step k+1 scans into the other record buffer
        self.overlap = bool(overlap) and self.world > 1
        # single GPU: the same flag pipelines the calls themselves -- step k+1 is enqueued (acgpu_match_device_begin)
        # before the count of step k is read back, so the GPU never waits for the host between steps
        self.pipeline = bool(overlap) and self.world == 1 and scan_fn is None and self.mode == MODE_ALL
        self._ticket = None
        self.outs = [torch.empty((self.cap, self.cols), dtype=torch.int32, device=device)
                     for _ in range(2 if (self.overlap or self.pipeline) else 1)]
        self.out = self.outs[0]
        self._slot = 0
        self._tmp = None  # Longest: records of the repair window
        self._k = 0
        self._pending = None  # (work, gathered, counts) of the all-gather still in flight
        self.scan_fn = scan_fn
        self.last_kernel = ""
        self.gathered = None
        self.counts = None
        self.chain_repairs = 0  # Longest: window re-runs of the last step (0 = speculation was right)
        self.chain_window = 4096  # Longest: first repair window in units (x4 until the chains meet)

    def own_ptr(self):
        return self.sb.own.data_ptr()

    def own_units_host(self, k=None):
        v = self.sb.own if k is None else self.sb.own[:k]
        return v.cpu().numpy().view(np.uint16)

    # ---- one native call on (part of) the shard --------------------------------------------------------------
    def _grow(self, which, need):
        cap = max(self.cap, int(need * 1.25) + 16)
        t = torch.empty((cap, self.cols), dtype=torch.int32, device=self.sb.buf.device)
        if which == "out":
            self.cap = cap
            self.out = self.outs[self._slot] = t
        else:
            self._tmp = t
        return t

    def _call(self, which, own_lo, own_hi, entry, profile=False):
        """Scan the owned sub-range [own_lo, own_hi) (shard-relative) of this rank's buffer into self.out / self._tmp.
        Records come back shard-relative.  Returns (n, chain_exit (shard-relative), profile dict | None)."""
        sb = self.sb
        first = self.rank == 0
        last = self.rank == self.world - 1
        v0 = sb.pad if first else 0  # the pad in front of rank 0's text is not part of the haystack
        v1 = sb.pad + sb.n_units + (0 if last else sb.right)
        shift = sb.pad - v0  # view position of shard position 0
        buf = self.out if which == "out" else self._tmp
        if buf is None:
            buf = self._grow(which, 1024)
        if self.scan_fn is not None:
            recs, ex = self.scan_fn(sb.buf[v0:v1].numpy().view(np.uint16), own_lo + shift, own_hi + shift, first, last,
                                    entry + shift)
            n = len(recs)
            if n > buf.shape[0]:
                buf = self._grow(which, n)
            if n:
                r = np.ascontiguousarray(recs[:, :self.cols], dtype=np.int32).copy()
                r[:, :2] -= shift
                buf[:n] = torch.from_numpy(r)
            return n, int(ex) - shift, None
        from . import _native as N
        stream = torch.cuda.current_stream().cuda_stream
        while True:
            n, rc, prof, ex = self.auto.match_device(sb.buf.data_ptr() + 2 * v0, v1 - v0, self.with_ids, buf.data_ptr(),
                                                     buf.shape[0], own=(own_lo + shift, own_hi + shift), text_begin=first,
                                                     text_end=last, chain_entry=entry + shift, stream=stream, profile=profile)
            if rc == N.E_OVERFLOW:
                buf = self._grow(which, n)
                continue
            N.check(rc, "acgpu_match_device")
            break
        if shift and n:
            buf[:n, :2] -= shift  # view-relative -> shard-relative
        if prof:
            self.last_kernel = prof["scan_kernel"]
        return n, ex - shift, prof

    def _scan(self, profile):
        if self.mode == MODE_LONGEST:
            return self._scan_longest(profile)
        if self.mode == MODE_SHORTEST:
            return self._scan_shortest(profile)
        n, _, prof = self._call("out", 0, self.sb.n_units, 0, profile)
        return n, prof

    # ---- Longest: speculative chain + one int64 down the ranks -----------------------------------------------
    def _chain_hop_recv(self):
        if self.rank == 0:
            return 0
        t = torch.empty(1, dtype=torch.int64, device=self.sb.buf.device)
        dist.recv(t, self.rank - 1, group=self.group)
        return int(t.item()) - self.rank * self.sb.n_units  # global -> shard-relative

    def _chain_hop_send(self, exit_pos):
        if self.rank + 1 < self.world:
            t = torch.tensor([exit_pos + self.rank * self.sb.n_units], dtype=torch.int64, device=self.sb.buf.device)
            dist.send(t, self.rank + 1, group=self.group)

    def _scan_longest(self, profile):
        n_own = self.sb.n_units
        n, ex, prof = self._call("out", 0, n_own, 0, profile)  # speculation: the chain enters at my first unit
        entry = self._chain_hop_recv()  # true entry >= 0: the previous rank's exit
        self.chain_repairs = 0
        if entry != 0:
            spec = self.out
            starts = spec[:n, 0].contiguous()
            w = int(self.chain_window)
            while True:
                w_end = min(w, n_own)
                self.chain_repairs += 1
                n_t, ex_t, _ = self._call("tmp", 0, w_end, entry)
                if w_end == n_own:  # the window is the whole shard: nothing of the speculation is kept
                    idx, ex_s = n, ex_t
                else:
                    # where the speculative chain leaves the window: max(w_end, end of its last match starting inside)
                    idx = int(torch.searchsorted(starts, torch.tensor([w_end], dtype=torch.int32, device=starts.device)).item())
                    ex_s = max(w_end, int(spec[idx - 1, 1].item())) if idx else w_end
                if w_end == n_own or ex_t == ex_s:
                    tail = spec[idx:n].clone()
                    if n_t + len(tail) > self.out.shape[0]:
                        self._grow("out", n_t + len(tail))
                    self.out[:n_t] = self._tmp[:n_t]
                    self.out[n_t:n_t + len(tail)] = tail
                    n = n_t + len(tail)
                    if w_end == n_own:
                        ex = ex_t
                    break
                w *= 4
        self._chain_hop_send(ex)
        return n, prof

    def _scan_shortest(self, profile):
        """Shortest: records are owned by their END; the chain state is the position of the last restart (the end of
        the last reported match, or what came in if this shard reported nothing).  Speculation: no restriction."""
        n_own, halo = self.sb.n_units, self.sb.halo
        none = -(self.sb.pad + 1)  # a restart position left of everything this rank can see restricts nothing
        n, _, prof = self._call("out", 0, n_own, none, profile)
        entry = max(self._chain_hop_recv(), none) if self.rank else none
        self.chain_repairs = 0
        if entry > -halo:  # a restart inside my halo can forbid matches that begin before it
            spec = self.out
            ends = spec[:n, 1].contiguous()
            w = int(self.chain_window)
            while True:
                w_end = min(w, n_own)
                self.chain_repairs += 1
                n_t, _, _ = self._call("tmp", 0, w_end, entry)
                # speculative records that end inside the window (ends ascend)
                idx = n if w_end == n_own else int(torch.searchsorted(
                    ends, torch.tensor([w_end], dtype=torch.int32, device=ends.device), right=True).item())
                last_t = int(self._tmp[n_t - 1, 1].item()) if n_t else entry
                last_s = int(spec[idx - 1, 1].item()) if idx else none
                floor = w_end - halo  # restart positions at or left of this restrict nothing that ends after the window
                if w_end == n_own or max(last_t, floor) == max(last_s, floor):
                    tail = spec[idx:n].clone()
                    if n_t + len(tail) > self.out.shape[0]:
                        self._grow("out", n_t + len(tail))
                    self.out[:n_t] = self._tmp[:n_t]
                    self.out[n_t:n_t + len(tail)] = tail
                    n = n_t + len(tail)
                    break
                w *= 4
        self._chain_hop_send(int(self.out[n - 1, 1].item()) if n else entry)
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
