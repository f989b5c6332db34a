"""Multi-GPU sharding of one long haystack: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" in the CPU tests -- and, host-staged, when several ranks have to share one GPU).  SURVEY.md 8e, one
row per matcher family:

* AhoCorasick (all matches) shards naturally: rank g owns a contiguous range of the haystack, needs the
  (max_keyword_len-1) units before it (left halo, received from rank g-1), starts at the root, and keeps the matches
  whose LAST unit it owns.
* WholeWord: a word belongs to the rank that owns its FIRST unit; left context 1 unit (is the previous unit a word
  character?), right halo max_keyword_len+1 units from rank g+1 (enough to see that a run is longer than any keyword).
* Longest: the lengths L[pos] need a right halo of max_keyword_len-1 units and are independent per shard; the greedy
  chain pos -> pos + max(L[pos],1) needs each shard's entry position = the previous shard's exit.  Every rank first
  runs its chain speculatively from its own first unit (all ranks in parallel); then the ranks ALL-GATHER their exits
  (one int64 each, all ranks at once -- not a hop down the ranks), a rank whose true entry differs re-runs a short window
  until the true chain leaves the window where the speculative one did -- from there on both are the same chain -- and
  the gather is repeated until it returns what the previous one did (two gathers unless a repair changed an exit).
* Shortest: a match belongs to the rank that owns its LAST unit (left halo as AhoCorasick); which occurrences are
  reported depends on where matching last restarted (the end of the previous reported match), handed down the ranks
  like the Longest chain position, with the same speculation (no restriction) and window repair.
* WholeWordLongest: a walk belongs to the rank that owns its FIRST unit (halos as WholeWord); which word starts the scan
  visits depends on where the previous rank's last walk stopped -- the same hop; a rank whose true entry lies inside its
  shard (the previous rank's last walk ran over the boundary) scans once more from there.

In all of them rank-local order is the reference's order, so the concatenation of the per-rank buffers by rank is the
reference's listener-call order for the whole haystack.

Data exchange, and what a step costs the host:
* halos are exchanged ONCE PER HAYSTACK (when the shard's text changes: haystack_changed()), not per step;
* every rank scans into a fixed-size GATHER BUFFER [16-byte header | gcap records]; the header is the
  acgpu_device_result {record count, redone flag} that the scan's last kernel writes in stream order
  (acgpu_shard.d_result), so ONE all-gather of the whole buffer moves counts and records together and nothing on the
  host sits between the scan and the collective.  gcap follows the largest count seen so far (with head-room); a step
  in which some rank's count exceeds it is detected by every rank from the gathered headers and redone by all of them
  with larger buffers (rare: the first step of a much denser haystack);
* AhoCorasick and WholeWord enqueue their scan with acgpu_match_device_begin and collect it with _end only after the
  gathered headers have arrived on the host: one blocking host synchronisation per step (the header read-back).  With
  overlap=True that read-back is the PREVIOUS step's, so the all-gather of step k runs under the scan of step k+1;
* Longest / Shortest / WholeWordLongest additionally pay the all-gathers of the chain exits (two per step as a rule,
  whatever the number of ranks) and the count read-back of their synchronous scan calls.
Positions stay shard-local int32 in the gathered buffer and become global int64 positions by adding
base[g] = g * units_per_rank (global_records()).
"""
import numpy as np
import torch
import torch.distributed as dist

from ._native import MODE_ALL, MODE_LONGEST, MODE_SHORTEST, MODE_WHOLEWORD, MODE_WWLONGEST

HDR = 4  # int32 words in front of the records of a gather buffer: acgpu_device_result {u64 n_records, u32 redone, u32 0}


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _host_staged(t, group=None):
    """gloo moves host memory: a device tensor is copied to the host, exchanged, and copied back (several ranks on one
    GPU -- RCCL cannot do that -- and the GPU tests of this module)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


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
    staged = _host_staged(sb.buf, group)
    dev = torch.device("cpu") if staged else sb.buf.device
    ops = []
    recv_l = recv_r = None
    # bytes on the wire: RCCL has no 16-bit integer type
    if sb.halo:
        if rank + 1 < world:
            ops.append(dist.P2POp(dist.isend, sb.tail_view().contiguous().view(torch.uint8).to(dev), rank + 1, group))
        if rank > 0:
            recv_l = torch.empty(2 * sb.halo, dtype=torch.uint8, device=dev)
            ops.append(dist.P2POp(dist.irecv, recv_l, rank - 1, group))
    if sb.right:
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, sb.head_view().contiguous().view(torch.uint8).to(dev), rank - 1, group))
        if rank + 1 < world:
            recv_r = torch.empty(2 * sb.right, dtype=torch.uint8, device=dev)
            ops.append(dist.P2POp(dist.irecv, recv_r, rank + 1, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if recv_l is not None:
        sb.halo_view().copy_(recv_l.view(torch.int16))
    if recv_r is not None:
        sb.right_view().copy_(recv_r.view(torch.int16))


def allgather_flat(out, send, group=None, async_op=False):
    """all_gather_into_tensor of equal-sized flat int32 buffers; host-staged under gloo on device tensors."""
    if _host_staged(send, group):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, send.cpu(), group=group)
        out.copy_(h_out)
        return None
    return dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)


def global_records(gathered, counts, units_per_rank, pad=0):
    """Concatenate per-rank records in rank order with global int64 positions (reference order of the whole text).
    Records in a gather buffer are relative to the rank's VIEW of its shard buffer: the first owned unit for rank 0,
    `pad` units in front of it (the aligned room of the left halo) for every other rank."""
    parts = []
    for g, n in enumerate(counts.tolist()):
        r = gathered[g, :n].to(torch.int64).clone()
        r[:, :2] += g * int(units_per_rank) - (int(pad) if g else 0)
        parts.append(r)
    return torch.cat(parts) if parts else torch.empty((0, gathered.shape[-1]), dtype=torch.int64)


class _Step:
    """One step in flight: the gather buffer it scans into, the gathered copy, and what collects it."""

    def __init__(self):
        self.gbuf = None       # (HDR + gcap*cols,) int32 on the device: [header | records]
        self.gcap = 0
        self.gathered = None   # (world, HDR + gcap*cols) int32
        self.hdr_host = None   # (world, HDR) int32, pinned when the buffers live on a GPU
        self.event = None      # recorded behind the header read-back
        self.work = None       # the all-gather still in flight (nccl, async)
        self.ticket = None     # acgpu_match_device_begin ticket (AhoCorasick, WholeWord)
        self.n = None          # local record count when the scan call was synchronous
        self.prof = None
        self.prof_on = False
        self.private_out = None  # a scan whose records did not fit gcap keeps them here (the step is then redone)


class ShardedMatcher:
    """One rank's end of the sharded match of one long haystack: device-resident shard, halo exchange (once per
    haystack), native scan into the gather buffer (acgpu_match_device[_begin]), [Longest/Shortest: chain hop],
    ONE all-gather of header + records.

    scan_fn (tests only) replaces the native scan so the plumbing can run under gloo on CPU; it has the contract of
    acgpu_match_device on one shard:
    scan_fn(view_units_np, own_begin, own_end, text_begin, text_end, chain_entry) -> ((n,cols) int32 records relative
    to the view, chain_exit)."""

    def __init__(self, automaton, n_units, with_ids=True, cap=1 << 20, device=None, group=None, scan_fn=None, halo=None,
                 overlap=False, mode=None, right_halo=None, adaptive=True, force_collectives=False):
        self.auto = automaton
        self.group = group
        self.rank, self.world = _world(group)
        # a world of ONE normally bypasses every collective; force_collectives keeps them (the gather buffers, the all-gather
        # left in flight under the next scan, the header read-back, the chain families' gathers of exits) -- how a one-GPU box
        # drives the whole multi-rank step over a real RCCL communicator
        self.collective = self.world > 1 or bool(force_collectives)
        self.with_ids = with_ids
        self.cols = 3 if with_ids else 2
        self.mode = automaton.mode if automaton is not None else (MODE_ALL if mode is None else mode)
        if halo is None or (right_halo is None and self.mode != MODE_ALL):
            max_len = automaton.info()["max_keyword_len"]
            if self.mode in (MODE_ALL, MODE_SHORTEST):
                halo, right_halo = max(0, max_len - 1), 0
            elif self.mode in (MODE_WHOLEWORD, MODE_WWLONGEST):
                halo, right_halo = 1, max_len + 1
            else:
                halo, right_halo = 0, max(0, max_len - 1)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if scan_fn is None else torch.device("cpu")
        self.device = torch.device(device)
        self.sb = ShardBuffer(n_units, halo, self.device, right_halo or 0)
        self.cap = int(cap)  # gcap: records per rank in the gather buffer
        self.adaptive = bool(adaptive)  # gcap shrinks to the largest count seen (+ head-room) after a step
        self._published = None
        self._pending_result = None  # result of a step that haystack_changed() completed
        # overlap: step() leaves its all-gather (world > 1) / its scan (world == 1) in flight and returns the PREVIOUS
        # step's result; the all-gather of step k then runs under the scan of step k+1
        self.overlap = bool(overlap)
        # AhoCorasick and WholeWord (fold-consistent word-character tables) have the asynchronous form of the native call; so
        # does LongestMatch where the chain entry is known when the scan is enqueued (one rank: it is the first unit)
        self.async_scan = scan_fn is None and (self.mode == MODE_ALL or (
            self.mode == MODE_WHOLEWORD and bool(automaton.info()["fold_consistent"])) or (
            self.mode == MODE_LONGEST and not self.collective))
        self.scan_fn = scan_fn
        self._k = 0
        self._inflight = None   # the _Step a previous step() left for the next one (overlap)
        self._free = []         # gather buffers of the current gcap, free for reuse
        self._halo_dirty = True
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._hop_device = self.device
        if self.world > 1 and self.device.type == "cuda" and dist.get_backend(group) == "gloo":
            self._hop_device = torch.device("cpu")
        self.out = None   # records view of the current step's gather buffer (or its private overflow buffer)
        self._cur = None
        self._tmp = None  # Longest / Shortest: records of the repair window
        self.last_kernel = ""
        self.gathered = None  # (world, gcap, cols) int32 view of the last COMPLETED step
        self.counts = None    # (world,) int64 on the host
        self.chain_repairs = 0  # Longest: window re-runs of the last step (0 = speculation was right)
        self.chain_entry_applied = None  # chain families: the entry the last step's records assume
        self.chain_window = 4096  # Longest: first repair window in units (x4 until the chains meet)
        self.host_syncs = 0   # blocking host synchronisations of the last step() on its own account (diagnostic)
        self.redone_steps = 0  # steps redone with larger gather buffers

    # ---- the shard's text ------------------------------------------------------------------------------------
    def own_ptr(self):
        return self.sb.own.data_ptr()

    def own_units_host(self, k=None):
        v = self.sb.own if k is None else self.sb.own[:k]
        return v.cpu().numpy().view(np.uint16)

    def haystack_changed(self):
        """Call BEFORE writing new text into sb.own (load() does): a step still in flight (overlap=True) is completed
        first -- it may have to be redone, which needs its text -- and the next step() exchanges the halos again.  The
        completed step's result dict is what that next step() returns."""
        if self._inflight is not None:
            st, self._inflight = self._inflight, None
            self._pending_result = self._collect(st)
        self._halo_dirty = True

    def load(self, units):
        """Copies this rank's shard (int16/uint16 tensor or numpy array of n_units code units) into the buffer."""
        if isinstance(units, np.ndarray):
            units = torch.from_numpy(np.ascontiguousarray(units).view(np.int16))
        self.haystack_changed()
        self.sb.own.copy_(units.view(torch.int16))

    # ---- gather buffers ----------------------------------------------------------------------------------------
    def _new_step(self):
        st = self._free.pop() if self._free else _Step()
        if st.gbuf is None or st.gcap != self.cap:
            st.gcap = self.cap
            # (only the header has to start as zeros: the records behind it are written by the scan and read up to its count --
            # zeroing all of a buffer sized for config 4's 84 M records was a 110 us fill kernel in front of a 540 us step)
            st.gbuf = torch.empty(HDR + self.cap * self.cols, dtype=torch.int32, device=self.device)
            st.gbuf[:HDR].zero_()
            if self.collective:
                st.gathered = torch.empty((self.world, HDR + self.cap * self.cols), dtype=torch.int32, device=self.device)
                st.hdr_host = torch.empty((self.world, HDR), dtype=torch.int32, pin_memory=self.device.type == "cuda")
            if self.device.type == "cuda":
                st.event = torch.cuda.Event()
        st.work = st.ticket = st.n = st.prof = st.private_out = None
        return st

    def _records(self, st):
        return st.gbuf[HDR:].view(st.gcap, self.cols)

    def _release(self, st):
        if st.gcap == self.cap and len(self._free) < 2:
            self._free.append(st)

    def _write_header(self, st, n):
        st.gbuf[:HDR].view(torch.int64).copy_(torch.tensor([int(n), 0], dtype=torch.int64))

    @property
    def shift(self):
        """view position of this rank's first owned unit (what record positions in its gather buffer are relative to)"""
        return 0 if self.rank == 0 else self.sb.pad

    # ---- one native call on (part of) the shard --------------------------------------------------------------
    def _grow(self, which, need):
        """A private record buffer: "tmp" (repair windows), or "out" when the step's records do not fit the gather
        buffer -- the step then finishes locally (the chain hop needs its exit) and is redone by all ranks."""
        cap = max(self.cap, int(need * 1.25) + 16)
        t = torch.empty((cap, self.cols), dtype=torch.int32, device=self.device)
        if which == "out":
            self.out = self._cur.private_out = t
        else:
            self._tmp = t
        return t

    def _view(self):
        sb = self.sb
        first = self.rank == 0
        last = self.rank == self.world - 1
        v0 = sb.pad if first else 0  # the pad in front of rank 0's text is not part of the haystack
        v1 = sb.pad + sb.n_units + (0 if last else sb.right)
        return v0, v1, first, last

    def _call(self, which, own_lo, own_hi, entry, profile=False, d_result=None):
        """Scan the owned sub-range [own_lo, own_hi) (shard-relative) of this rank's buffer into self.out / self._tmp.
        Records come back relative to the VIEW (see global_records); own_lo, own_hi, entry and the returned chain exit
        are shard-relative.  Returns (n, chain_exit, profile dict | None)."""
        sb = self.sb
        v0, v1, first, last = self._view()
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
                buf[:n] = torch.from_numpy(np.ascontiguousarray(recs[:, :self.cols], dtype=np.int32))
            return n, int(ex) - shift, None
        from . import _native as N
        stream = torch.cuda.current_stream().cuda_stream
        while True:
            n, rc, prof, ex = self.auto.match_device(sb.buf.data_ptr() + 2 * v0, v1 - v0, self.with_ids, buf.data_ptr(),
                                                     buf.shape[0], own=(own_lo + shift, own_hi + shift), text_begin=first,
                                                     text_end=last, chain_entry=entry + shift, stream=stream, profile=profile,
                                                     d_result=d_result)
            self.host_syncs += 1
            if rc == N.E_OVERFLOW:
                buf = self._grow(which, n)
                d_result = None  # (the header already says n > gcap)
                continue
            N.check(rc, "acgpu_match_device")
            break
        if prof:
            self.last_kernel = prof["scan_kernel"]
        return n, ex - shift, prof

    def _scan(self, profile, st):
        """Synchronous scan of the whole shard into st's gather buffer; the header is in place when it returns."""
        self._cur = st
        self.out = self._records(st)
        if self.mode == MODE_LONGEST:
            n, prof = self._scan_longest(profile)
        elif self.mode == MODE_SHORTEST:
            n, prof = self._scan_shortest(profile)
        elif self.mode == MODE_WWLONGEST:
            n, prof = self._scan_wwlongest(profile)
        else:
            native = self.scan_fn is None
            n, _, prof = self._call("out", 0, self.sb.n_units, 0, profile, d_result=st.gbuf.data_ptr() if native else None)
            if native:
                return n, prof
        self._write_header(st, n)
        return n, prof

    # ---- the chain families: speculation on every rank + all-gathers of the chain exits until they settle ------------------
    # Longest / Shortest / WholeWordLongest: what a rank reports depends on one number from the rank before it (where the
    # greedy chain enters / where matching last restarted / from where the scan looks for its next word start).  Round 1's
    # driver sent that number DOWN the ranks: world-1 serial hops, each behind the sender's scan and repair.  Here every rank
    # scans speculatively at once ("nothing comes in"), then the ranks all-gather their exits -- one small collective, all
    # ranks in parallel -- take the exit of the rank before them as their entry, repair the head of their shard if that
    # entry differs from the assumption, and gather again: the loop ends when a gather returns what the previous one did.
    # A repair almost never changes a rank's exit (the chains merge within a few keyword lengths), so a step costs two
    # gathers however many ranks there are; the worst case -- every exit depends on the entry -- is the old serial hop.
    _NONE = -(1 << 62)  # "nothing comes in" in global coordinates

    def _gather_exits(self, ex_global):
        """All ranks' exits (global positions), identical on every rank."""
        t = torch.tensor([int(ex_global)], dtype=torch.int64, device=self._hop_device)
        out = torch.empty(self.world, dtype=torch.int64, device=self._hop_device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        self.host_syncs += 1
        return out.cpu().tolist()

    def _scan_chain(self, profile, spec, repair, to_global, from_global):
        """spec() -> (n, exit) under the assumption that nothing comes in; repair(entry) -> (n, exit) for a true entry (always
        from the speculative records, which spec() left in self.out / keeps in self._spec); to_global / from_global map a
        rank's exit to and from the number that travels."""
        n, ex, prof = spec(profile)
        self.chain_repairs = 0
        if not self.collective:
            return n, prof
        applied = None   # the entry my current records assume (None: the speculation)
        prev = None
        self._spec_dirty = False
        while True:
            exits = self._gather_exits(to_global(ex, n, applied))
            if exits == prev:
                break
            prev = exits
            entry = from_global(exits[self.rank - 1]) if self.rank else None
            if entry != applied:
                if self._spec_dirty:  # a second repair starts from the speculation again
                    n, ex, _ = spec(False)
                    self._spec_dirty = False
                if entry is not None:
                    n, ex = repair(entry, n, ex)
                applied = entry
        self.chain_entry_applied = applied  # (shard-relative; None on rank 0: diagnostic, bench.py's self-check)
        return n, prof

    def _scan_longest(self, profile):
        n_own = self.sb.n_units
        base = self.rank * n_own

        def spec(prof_on):
            n, ex, prof = self._call("out", 0, n_own, 0, prof_on)  # the chain enters at my first unit
            return n, ex, prof

        def repair(entry, n, ex):
            if entry == 0:
                return n, ex
            spec_recs, shift = self.out, self.shift  # (records are view-relative, the chain positions shard-relative)
            starts = spec_recs[:n, 0].contiguous()
            w = int(self.chain_window)
            while True:
                w_end = min(w, n_own)
                self.chain_repairs += 1
                n_t, ex_t, _ = self._call("tmp", 0, w_end, entry)
                if w_end == n_own:  # the window is the whole shard: nothing of the speculation is kept
                    idx, ex_s = n, ex_t
                else:
                    # where the speculative chain leaves the window: max(w_end, end of its last match starting inside) -- index
                    # and end computed on the device, ONE read-back
                    idx_t = torch.searchsorted(starts, torch.tensor([w_end + shift], dtype=torch.int32, device=starts.device))
                    end_t = spec_recs[(idx_t - 1).clamp(min=0), 1]
                    idx, last_end = torch.stack([idx_t.view(()).to(torch.int64), end_t.view(()).to(torch.int64)]).cpu().tolist()
                    ex_s = max(w_end, last_end - shift) if idx else w_end
                if w_end == n_own or ex_t == ex_s:
                    tail = spec_recs[idx:n].clone()
                    if n_t + len(tail) > self.out.shape[0]:
                        self._grow("out", n_t + len(tail))
                    self.out[:n_t] = self._tmp[:n_t]
                    self.out[n_t:n_t + len(tail)] = tail
                    self._spec_dirty = True
                    return n_t + len(tail), (ex_t if w_end == n_own else ex)
                w *= 4

        return self._scan_chain(profile, spec, repair, lambda ex, n, applied: ex + base,
                                lambda g: max(g - base, 0))

    def _scan_wwlongest(self, profile):
        """WholeWordLongest: speculation = the scan enters at my first unit; the true entry is the position behind the stop of
        the previous rank's last walk.  Walks are short (at most max_keyword_len units), so an entry inside my shard lies
        within my first few units: the shard is scanned once more from there."""
        n_own = self.sb.n_units
        base = self.rank * n_own

        def spec(prof_on):
            return self._call("out", 0, n_own, 0, prof_on)

        def repair(entry, n, ex):
            if entry <= 0:
                return n, ex
            self.chain_repairs += 1
            self._spec_dirty = True
            n2, ex2, _ = self._call("out", 0, n_own, entry)
            return n2, max(ex2, entry)

        return self._scan_chain(profile, spec, repair, lambda ex, n, applied: ex + base,
                                lambda g: max(g - base, 0))

    def _scan_shortest(self, profile):
        """Shortest: records are owned by their END; the chain state is the position of the last restart (the end of the last
        reported match, or what came in if this shard reported nothing).  Speculation: no restriction."""
        n_own, halo = self.sb.n_units, self.sb.halo
        base = self.rank * n_own
        none = -(self.sb.pad + 1)  # a restart position left of everything this rank can see restricts nothing
        shift = self.shift  # (records are view-relative, the restart positions shard-relative)

        def spec(prof_on):
            n, _, prof = self._call("out", 0, n_own, none, prof_on)
            return n, None, prof

        def repair(entry, n, ex):
            if entry <= -halo:  # a restart left of my halo forbids nothing
                return n, ex
            spec_recs = self.out
            ends = spec_recs[:n, 1].contiguous()
            w = int(self.chain_window)
            while True:
                w_end = min(w, n_own)
                self.chain_repairs += 1
                n_t, _, _ = self._call("tmp", 0, w_end, entry)
                # speculative records that end inside the window (ends ascend); index and both last ends in ONE read-back
                if w_end == n_own:
                    idx_t = torch.tensor([n], dtype=torch.int64, device=ends.device)
                else:
                    idx_t = torch.searchsorted(ends, torch.tensor([w_end + shift], dtype=torch.int32, device=ends.device), right=True)
                ls_t = spec_recs[(idx_t - 1).clamp(min=0), 1] if n else torch.zeros(1, dtype=torch.int32, device=ends.device)
                lt_t = self._tmp[max(n_t - 1, 0):max(n_t - 1, 0) + 1, 1]
                idx, ls, lt = torch.stack([idx_t.view(()).to(torch.int64), ls_t.view(()).to(torch.int64), lt_t.view(()).to(torch.int64)]).cpu().tolist()
                last_t = lt - shift if n_t else entry
                last_s = ls - shift if idx else none
                floor = w_end - halo  # restart positions at or left of this restrict nothing that ends after the window
                if w_end == n_own or max(last_t, floor) == max(last_s, floor):
                    tail = spec_recs[idx:n].clone()
                    if n_t + len(tail) > self.out.shape[0]:
                        self._grow("out", n_t + len(tail))
                    self.out[:n_t] = self._tmp[:n_t]
                    self.out[n_t:n_t + len(tail)] = tail
                    self._spec_dirty = True
                    return n_t + len(tail), None
                w *= 4

        def to_global(ex, n, applied):  # the end of my last reported match, or what came in
            if n:
                return int(self.out[n - 1, 1].item()) - shift + base
            return self._NONE if applied is None else applied + base

        return self._scan_chain(profile, spec, repair, to_global,
                                lambda g: none if g == self._NONE else max(g - base, none))

    # ---- a step: enqueue (scan + all-gather + header read-back), collect ----------------------------------------
    def _enqueue(self, profile):
        from . import _native as N
        st = self._new_step()
        st.prof_on = profile
        if self.async_scan:
            sb = self.sb
            v0, v1, first, last = self._view()
            shift = sb.pad - v0
            tk, rc = self.auto.match_device_begin(sb.buf.data_ptr() + 2 * v0, v1 - v0, self.with_ids,
                                                  st.gbuf.data_ptr() + 4 * HDR, st.gcap, own=(shift, shift + sb.n_units),
                                                  text_begin=first, text_end=last,
                                                  stream=torch.cuda.current_stream().cuda_stream, profile=profile,
                                                  d_result=st.gbuf.data_ptr())
            N.check(rc, "acgpu_match_device_begin")
            st.ticket = tk
        else:
            st.n, st.prof = self._scan(profile, st)
        if self.collective:
            if self.device.type == "cuda" and not _host_staged(st.gbuf, self.group):
                st.work = allgather_flat(st.gathered.view(-1), st.gbuf, self.group, async_op=True)
                with torch.cuda.stream(self._side):
                    st.work.wait()  # (the side stream waits for RCCL's; the scan stream is free for the next step)
                    st.hdr_host.copy_(st.gathered[:, :HDR], non_blocking=True)
                    st.event.record(self._side)
            else:
                allgather_flat(st.gathered.view(-1), st.gbuf, self.group)
                st.hdr_host.copy_(st.gathered[:, :HDR])
        return st

    def _collect(self, st):
        """Waits for what _enqueue left in flight (ONE blocking host synchronisation: the header read-back / the scan's
        done event), publishes gathered/counts, and returns the step's result dict.  A step whose records did not fit
        the gather buffers on some rank -- every rank sees that in the gathered headers -- is redone by all ranks."""
        from . import _native as N
        n, prof = st.n, st.prof
        if self.collective and st.work is not None:
            st.event.synchronize()
            self.host_syncs += 1
        if self.collective:
            hdr = st.hdr_host.numpy()
            counts = (hdr[:, 0].astype(np.int64) & 0xffffffff) | (hdr[:, 1].astype(np.int64) << 32)
            bad = bool((counts > st.gcap).any() or (hdr[:, 2] != 0).any())
        if st.ticket is not None:
            tk, st.ticket = st.ticket, None
            if self.collective and bad:
                # every rank redoes this step anyway (_redo): the ticket is given up, not collected -- collecting it would
                # make the ranks whose own scan has to be redone scan twice, and leave the others waiting for them
                N.check(self.auto.match_device_abandon(tk), "acgpu_match_device_abandon")
            else:
                n, rc, prof = self.auto.match_device_end(tk, profile=st.prof_on)  # (world > 1: its event has long passed)
                if not self.collective:
                    self.host_syncs += 1
                if rc != N.E_OVERFLOW:
                    N.check(rc, "acgpu_match_device_end")
        if not self.collective:
            counts = np.array([n], dtype=np.int64)
            bad = n > st.gcap
        if bad:
            return self._redo(st, int(counts.max()), counts)
        if prof:
            self.last_kernel = prof["scan_kernel"]
        if self.collective:
            self.gathered = st.gathered[:, HDR:].view(self.world, st.gcap, self.cols)
        else:
            self.gathered = self._records(st)[:n].unsqueeze(0)
        self.counts = counts
        # the published step's buffers back `gathered` until the next step is published; then they are free again
        old, self._published = self._published, st
        if old is not None:
            self._release(old)
        if self.adaptive:  # the next gather buffers follow the largest count (the all-gather moves gcap records per rank)
            target = int(int(counts.max()) * 1.0625) + 1024
            if target < 0.8 * self.cap:
                self.cap = target
                self._free = []
        r = {"n_local": int(counts[self.rank]), "n_total": int(counts.sum()), "scan_ms": 0.0, "finalize_ms": 0.0}
        if prof:
            r.update(scan_ms=prof["scan_ms"], finalize_ms=prof["finalize_ms"])
        return r

    def _redo(self, st, need, counts):
        """All ranks arrive here together (the decision comes from the gathered headers): larger gather buffers, then the
        step once more, synchronously.  A later step that is already in flight keeps its own (old) buffers and is
        collected -- or redone -- on its own account."""
        self.redone_steps += 1
        self.cap = max(self.cap, int(need * 1.25) + 16)
        self._free = []
        prof_on = st.prof_on
        st2 = self._new_step()
        st2.prof_on = prof_on
        async_scan, self.async_scan = self.async_scan, False
        try:
            if self.scan_fn is None and self.mode == MODE_ALL:
                self._cur = st2
                self.out = self._records(st2)
                st2.n, _, st2.prof = self._call("out", 0, self.sb.n_units, 0, prof_on, d_result=st2.gbuf.data_ptr())
            else:
                st2.n, st2.prof = self._scan(prof_on, st2)
            if self.collective:
                if self.device.type == "cuda":
                    torch.cuda.current_stream().synchronize()
                allgather_flat(st2.gathered.view(-1), st2.gbuf, self.group)
                st2.hdr_host.copy_(st2.gathered[:, :HDR])
        finally:
            self.async_scan = async_scan
        return self._collect(st2)

    def step(self, profile=False):
        """[halo exchange, first step of a haystack] -> scan -> all-gather.  Returns a dict with n_local, n_total and
        (profile) kernel timings.  With overlap=True the step is left in flight and the PREVIOUS step's dict is returned
        (None for the first call); `gathered`/`counts` describe the last COMPLETED step until finish() is called."""
        self.host_syncs = 0
        if self._halo_dirty:
            exchange_halo(self.sb, self.group)
            self._halo_dirty = False
        self._k += 1
        st = self._enqueue(profile)
        if not self.overlap:
            return self._collect(st)
        prev, self._inflight = self._inflight, st
        if prev is not None:
            return self._collect(prev)
        r, self._pending_result = self._pending_result, None
        return r

    def finish(self):
        """Completes what step() left in flight (overlap=True) and returns that step's result dict."""
        st, self._inflight = self._inflight, None
        if st is not None:
            return self._collect(st)
        r, self._pending_result = self._pending_result, None
        return r

    def global_records(self):
        return global_records(self.gathered, self.counts, self.sb.n_units, self.sb.pad)
