"""Clip partitioning across the GPUs of one MI355X node -- new work, the reference is single-GPU.

Layout for N ranks (N in 1, 2, 4, 8, ...):  2-way CFG x (N/2)-way frames.
    rank r:  half = r // frame_ranks  (0 = unconditional, 1 = conditional)   [N == 1: both halves on the one rank]
             shard = r %  frame_ranks -> frames [f0, f1) of the clip (contiguous, sizes differ by at most 1)
Weights are replicated.  Exchanges inside one denoise step (RCCL over xGMI through torch.distributed):
    * temporal GroupNorm + (3,1,1) convolution (twice per TemporalResnetBlock), ONE exchange group each:
        - the RAW boundary frames of the block's input go to the neighbour shards asynchronously (batched p2p on the "data"
          communicator) as soon as that input exists,
        - the shard's GroupNorm partials (a few KB) are all-gathered on the "ctl" communicator -- the only point of the group
          the compute stream waits for --, every rank combines all partials itself in entry order (bit-identical statistics
          on every rank, no reduction-order dependence) and normalises its own frames,
        - the convolution of the INTERIOR frames runs while the boundary frames travel; they are normalised on arrival with
          the same statistics and the two boundary frames are convolved last.
    * temporal self-attention: ONE all_gather_into_tensor of the normed hidden tokens (C columns; r02 gathered K|V = 2C) into
      a preallocated [frame_ranks x T_max frames] buffer (the LayerNorm writes this rank's slot in place, the collective runs
      asynchronously under the Q projection, every rank then projects K|V for all key slots, and the attention kernel masks
      the padding frames of uneven shards -- no pad / concat / compaction copies)                   (frame group)
    * CFG combine: the two halves of one frame shard swap their noise predictions                 (pair group)
and once per clip: all-gather of the final latents before the VAE decode, whose chunks are independent and are
dealt round-robin to ALL ranks.  Everything per-frame (2-D convs, spatial norms/attention, FFs, the adapter warps,
zero convs, the Euler step) needs no communication.

With two networks of a step in flight (adapter trunk || UNet encoder on two HIP streams) ONE host thread enqueues both, layer
by layer in lockstep (blocks.run_lockstep): every rank runs the same program, so every communicator sees the same sequence of
collectives on every rank -- the precondition for RCCL not to deadlock -- by construction, with no thread, token or timing
involved; a network's wait for its partials / halo frames / token gather is a STREAM wait that the other network's kernels
cover, the host never blocks on the transport.

``Comm`` implementations: ``TorchComm`` (torch.distributed: "nccl" = RCCL on the GPUs, "gloo" in the CPU tests) and
``ThreadComm`` (virtual ranks as threads of one process -- lets the whole sharded HIP path be checked against the
unsharded one on a single GPU).
"""
import os
import threading

import torch


def split_frames(T, n):
    """contiguous shards, larger ones first: 25 over 4 -> [(0,7),(7,13),(13,19),(19,25)]"""
    base, extra = divmod(T, n)
    out, f = [], 0
    for i in range(n):
        k = base + (1 if i < extra else 0)
        out.append((f, f + k))
        f += k
    return out


class Layout:
    def __init__(self, world, rank, T, cfg_ranks=None, base=0):
        """cfg_ranks: None = 2 whenever world >= 2 (the product layout).  1 = frames only (both CFG halves on every
        rank) -- used by the exchange-primitive tests to shard frames over 2 ranks; the pipelines use the default.
        base: this layout covers the GLOBAL ranks [base, base + world) (one group of GroupedWindowParallel); ``rank`` is the
        position inside it, the group lists hold global ranks."""
        if cfg_ranks is None:
            assert world >= 1 and (world == 1 or world % 2 == 0), "1 or an even number of ranks"
            cfg_ranks = 2 if world >= 2 else 1
        assert cfg_ranks in (1, 2) and world % cfg_ranks == 0
        self.world, self.rank, self.T = world, rank, T
        self.cfg_ranks = cfg_ranks
        self.frame_ranks = world // self.cfg_ranks
        assert self.frame_ranks <= T
        self.half = rank // self.frame_ranks if self.cfg_ranks == 2 else None
        self.shard = rank % self.frame_ranks
        self.bounds = split_frames(T, self.frame_ranks)
        self.f0, self.f1 = self.bounds[self.shard]
        self.T_loc = self.f1 - self.f0
        self.T_max = self.bounds[0][1] - self.bounds[0][0]
        h = self.half or 0
        self.base = base
        self.frame_group = [base + h * self.frame_ranks + s for s in range(self.frame_ranks)]
        self.pair_group = [base + self.shard, base + self.frame_ranks + self.shard] if self.cfg_ranks == 2 else [base + rank]
        self.prev_rank = self.frame_group[self.shard - 1] if self.shard > 0 else None
        self.next_rank = self.frame_group[self.shard + 1] if self.shard + 1 < self.frame_ranks else None
        self.B_loc = 1 if self.cfg_ranks == 2 else 2

    @property
    def sharded_frames(self):
        return self.frame_ranks > 1


# ---------------------------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed backend (process group must be initialised).  Sub-groups are created once, by every rank,
    in the same order (a torch.distributed requirement)."""

    # RCCL has never carried the two-network lockstep order on real links (every SCALE record so far is a skipped one): the
    # single-stream order is the default on this transport; MOFA_SHARD_TWO_STREAMS=1 (the same on every rank) opts in
    two_streams_default = os.environ.get("MOFA_SHARD_TWO_STREAMS", "0") == "1"

    def __init__(self, layout_of_rank, two_lanes=None):
        import torch.distributed as dist
        self.dist = dist
        two_lanes = self.two_streams_default if two_lanes is None else two_lanes
        self.rank = dist.get_rank()
        world = dist.get_world_size()
        self._groups = {}
        seen = []
        for r in range(world):
            lay = layout_of_rank(r)
            for g in (tuple(lay.frame_group), tuple(lay.pair_group)):
                if g not in seen:
                    seen.append(g)
        # per group of ranks: "" = bulk (token gather, final latents, CFG pair), "ctl" = GroupNorm partials, "data" = halo
        # frames -- a small latency-bound exchange never queues behind a bulk transfer.  With ``two_lanes`` (the two networks
        # of a step enqueued in lockstep on two HIP streams, FrameParallel.two_streams) the step's second network gets its OWN
        # three communicators, so that no exchange of one network queues behind the other network's stream (round-4 advice:
        # lane 1's partials gather on a shared "ctl" communicator would wait for lane 0's compute stream).  Created by every
        # rank in the same order (a torch.distributed requirement); ``two_lanes`` must therefore be the same on every rank.
        self.two_lanes = two_lanes
        self._lanes = []
        for lane in range(2 if two_lanes else 1):
            comms = {}
            for kind in ("bulk", "ctl", "data"):
                comms[kind] = {g: (dist.new_group(list(g)) if len(g) > 1 else None) for g in seen}
            self._lanes.append(comms)
        self._groups = self._lanes[0]["bulk"]
        self._world_group = None

    def _g(self, ranks):
        return self._groups[tuple(ranks)]

    def _lane(self, lane):
        """the communicator set of a network lane (lane 1 falls back to lane 0's when the second set was not created)"""
        return self._lanes[lane if lane < len(self._lanes) else 0]

    def all_reduce_sum(self, t, ranks):
        if len(ranks) > 1:
            self.dist.all_reduce(t, group=self._g(ranks))
        return t

    def all_gather(self, t, ranks):
        if len(ranks) == 1:
            return [t]
        outs = [torch.empty_like(t) for _ in ranks]
        self.dist.all_gather(outs, t.contiguous(), group=self._g(ranks))
        return outs

    def all_gather_world(self, t):
        n = self.dist.get_world_size()
        if n == 1:
            return [t]
        t = t.contiguous()
        flat = torch.empty((n * t.numel(),), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(flat, t.reshape(-1))
        return list(flat.reshape((n,) + tuple(t.shape)).unbind(0))

    def all_gather_into(self, buf, slot_rows, ranks, lane=0):
        """In-place all-gather: ``buf`` [len(ranks) * slot_rows, C] already holds this rank's rows in its own slot; after
        ``wait()`` it holds every rank's.  Asynchronous: RCCL runs the collective on its own stream, kernels launched
        between this call and ``wait()`` overlap it (``wait`` makes the CURRENT stream wait, the host does not block)."""
        if len(ranks) == 1:
            return _Done()
        i = list(ranks).index(self.rank)
        grp = self._lane(lane)["bulk"][tuple(ranks)]
        return self.dist.all_gather_into_tensor(buf, buf[i * slot_rows:(i + 1) * slot_rows], group=grp, async_op=True)

    def gather_small_into(self, buf, slot_rows, ranks, lane=0):
        """the in-place all-gather of ``all_gather_into`` on the lane's "ctl" communicator, complete in stream order when it
        returns (the current stream waits for it; the host does not block on RCCL)"""
        if len(ranks) > 1:
            i = list(ranks).index(self.rank)
            self.dist.all_gather_into_tensor(buf, buf[i * slot_rows:(i + 1) * slot_rows], group=self._lane(lane)["ctl"][tuple(ranks)])
        return buf

    def halo_begin(self, first, last, prev_rank, next_rank, ranks, lane=0):
        """send `first` to prev and `last` to next; receive prev's last and next's first (None at the clip ends).
        Asynchronous (the lane's "data" communicator's stream): ``wait()`` -> (from_prev, from_next) makes the current stream wait."""
        dist = self.dist
        grp = self._lane(lane)["data"][tuple(ranks)]
        ops, from_prev, from_next = [], None, None
        if prev_rank is not None:
            from_prev = torch.empty_like(last)
            ops += [dist.P2POp(dist.isend, first.contiguous(), prev_rank, group=grp), dist.P2POp(dist.irecv, from_prev, prev_rank, group=grp)]
        if next_rank is not None:
            from_next = torch.empty_like(first)
            ops += [dist.P2POp(dist.isend, last.contiguous(), next_rank, group=grp), dist.P2POp(dist.irecv, from_next, next_rank, group=grp)]
        return _HaloWork(dist.batch_isend_irecv(ops) if ops else [], from_prev, from_next, (first, last))


class _Done:
    def wait(self):
        return True


class _HaloWork:
    def __init__(self, works, from_prev, from_next, keep=None):
        self.works, self.res, self.keep = works, (from_prev, from_next), keep   # keep: the send buffers stay alive until wait()

    def wait(self):
        for w in self.works:
            w.wait()
        self.works, self.keep = [], None
        return self.res


class ThreadWorld:
    """shared state of N virtual ranks living in N threads of one process"""

    def __init__(self, world):
        self.world = world
        self.lock = threading.Lock()
        self.barriers = {}
        self.slots = {}

    def _group_state(self, ranks):
        key = tuple(ranks)
        with self.lock:
            if key not in self.barriers:
                self.barriers[key] = threading.Barrier(len(key))
                self.slots[key] = [None] * len(key)
        return self.barriers[key], self.slots[key]


class ThreadComm:
    """Virtual ranks as threads of one process on ONE GPU (the sharded HIP path checked against the unsharded one without a
    multi-GPU node).  The ranks' launches may sit on different HIP streams (every rank's second network has a stream of its
    own), so an exchange is: drain my stream, publish, barrier, copy the peers' tensors on my stream, drain, barrier -- host
    synchronous throughout, which is fine for a checker transport and makes every hand-over race free."""

    def __init__(self, tworld, rank):
        self.tw, self.rank = tworld, rank

    @staticmethod
    def _drain(t):
        ts = t if isinstance(t, (tuple, list)) else (t,)
        if any(x is not None and x.is_cuda for x in ts):
            torch.cuda.current_stream().synchronize()

    def _exchange(self, t, ranks):
        """-> the tensors (or tuples of tensors) every rank of ``ranks`` published, in rank order, as private copies"""
        bar, slots = self.tw._group_state(ranks)
        i = list(ranks).index(self.rank)
        self._drain(t)
        slots[i] = t
        bar.wait()
        cp = lambda x: None if x is None else x.clone()       # noqa: E731
        got = [(tuple(cp(x) for x in g) if isinstance(g, (tuple, list)) else cp(g)) if j != i else g for j, g in enumerate(slots)]
        self._drain(t)
        bar.wait()
        return got

    def all_reduce_sum(self, t, ranks):
        if len(ranks) == 1:
            return t
        got = self._exchange(t.clone(), ranks)
        acc = got[0].clone()
        for g in got[1:]:
            acc = acc + g          # fixed rank order on every rank -> identical result everywhere
        t.copy_(acc)
        return t

    def all_gather(self, t, ranks):
        if len(ranks) == 1:
            return [t]
        return self._exchange(t.contiguous(), ranks)

    def all_gather_world(self, t):
        return self.all_gather(t, list(range(self.tw.world)))

    def all_gather_into(self, buf, slot_rows, ranks, lane=0):
        if len(ranks) == 1:
            return _Done()
        i = list(ranks).index(self.rank)
        got = self._exchange(buf[i * slot_rows:(i + 1) * slot_rows], ranks)
        for j, g in enumerate(got):
            if j != i:
                buf[j * slot_rows:(j + 1) * slot_rows].copy_(g)
        return _Done()

    def gather_small_into(self, buf, slot_rows, ranks, lane=0):
        self.all_gather_into(buf, slot_rows, ranks)
        return buf

    def halo_begin(self, first, last, prev_rank, next_rank, ranks, lane=0):
        if len(ranks) == 1:
            return _HaloWork([], None, None)
        got = self._exchange((first, last), ranks)
        ranks = list(ranks)
        from_prev = got[ranks.index(prev_rank)][1] if prev_rank is not None else None
        from_next = got[ranks.index(next_rank)][0] if next_rank is not None else None
        return _HaloWork([], from_prev, from_next)


# ---------------------------------------------------------------------------------------------------------
class FrameParallel:
    """The per-rank object blocks consult (``Ctx.par``) when a clip's frames are sharded."""

    def __init__(self, layout, comm, p2p=True):
        self.lay, self.comm = layout, comm
        self.T_full, self.T_loc, self.f0, self.f1 = layout.T, layout.T_loc, layout.f0, layout.f1
        self.p2p = p2p
        self.kv_inplace = True      # temporal attention K|V: in-place asynchronous all_gather_into_tensor (else: compacting gather)
        self.gather_hidden = True   # ... of the normed hidden tokens (C columns; K|V projected after the gather) instead of K|V (2C)
        # adapter trunk || UNet encoder on two HIP streams, enqueued layer by layer in lockstep: the transport's default (on for
        # the in-process ThreadComm; OFF for TorchComm until an N-GPU RCCL run has been recorded, MOFA_SHARD_TWO_STREAMS=1)
        # Derived from THIS communicator (round-5 advice): a TorchComm built with one communicator set per group must never run
        # the lockstep order (both networks would share bulk / ctl / data), one built with two lanes runs it.
        self.two_streams = bool(getattr(comm, "two_lanes", getattr(comm, "two_streams_default", True)))
        self.split_convs = False    # (3,1,1) convolutions as interior + boundary launches while the halo frames travel: OFF --
                                    # on the 1-GPU proxy of a rank of 8 the extra launches cost 3.1 ms of a 49 ms step while
                                    # all halo frames of a step are <= 2.5 ms of wire time that the second network already
                                    # covers (profiles/r04_shard_proxy.log); kept for links slower than xGMI
        self.lane = 0               # which network of the step is being enqueued (0 / 1): selects the bulk communicator and
                                    # the cached exchange buffers, so the two networks never queue behind each other
        self._part_bufs = {}
        self.log = None             # tests: list that receives (lane, kind) of every exchange issued

    def self_check(self, device):
        """Runs the two transport-specific fast paths once on small known data -- the in-place asynchronous
        all_gather_into_tensor whose input aliases its output slot (K|V exchange) and the batched isend / irecv halo
        exchange -- and compares with what the plain list all_gather delivers.  A path that raises or returns wrong data is
        switched off ON EVERY RANK of the frame group (the verdict is all-reduced), so the run continues on the conservative
        path (pad + all_gather + compaction, as for more than 32 key slots; all_gather-based halo).  Covers WRONG-DATA failures
        and exceptions every rank of the group raises alike (an unsupported call); a rank that raises alone while its peers
        are already inside the collective leaves them blocked -- that case ends at the process group's own
        timeout (torch.distributed's default), not in this check.  Returns a dict for the caller to report.  Meant for the
        first run on a new transport: bench.py calls it in shard mode."""
        lay, grp = self.lay, self.lay.frame_group
        device = torch.device(device)
        report = {"kv_gather": "in-place all_gather_into_tensor", "halo": "batched p2p" if self.p2p else "all_gather"}
        if len(grp) == 1:
            return report
        rows, C = 8, 16

        def pattern(shard, nrows):
            """row r of a shard holds shard * 128 + r in every column: a slot landing at the wrong offset, a shifted slot or
            rows permuted inside a slot all change the comparison (fp16 holds integers below 2048 exactly; r < 64)"""
            col = torch.arange(nrows, dtype=torch.float32, device=device) + 128.0 * shard
            return col.to(torch.float16).unsqueeze(1).repeat(1, C).contiguous()
        mine = pattern(lay.shard, lay.T_loc * rows)
        ref = self.comm.all_gather(pattern(lay.shard, lay.T_max * rows), grp)
        # --- K|V path
        ok = 1.0
        try:
            buf, own = self.kv_buffer(rows, C, device)
            buf.zero_()
            own.copy_(mine)
            self.kv_gather_begin(buf, rows).wait()
            slot = lay.T_max * rows
            for s_, (a, b) in enumerate(lay.bounds):
                n = (b - a) * rows
                if not torch.equal(buf[s_ * slot:s_ * slot + n], ref[s_][:n]):
                    ok = 0.0
        except Exception as e:  # noqa: BLE001
            ok = 0.0
            report["kv_gather_error"] = repr(e)[:200]
        flag = torch.tensor([ok], dtype=torch.float64, device=device)
        self.comm.all_reduce_sum(flag, grp)
        if float(flag.item()) < len(grp):
            self.kv_inplace = False
            report["kv_gather"] = "compacting all_gather (the in-place path failed its self-check)"
        # --- halo path
        if self.p2p:
            ok = 1.0
            try:
                first, last = pattern(lay.shard, rows), pattern(lay.shard, rows) + 1024.0
                fp, fn = self.comm.halo_begin(first, last, lay.prev_rank, lay.next_rank, grp).wait()
                if lay.prev_rank is not None and not torch.equal(fp, pattern(lay.shard - 1, rows) + 1024.0):
                    ok = 0.0
                if lay.next_rank is not None and not torch.equal(fn, pattern(lay.shard + 1, rows)):
                    ok = 0.0
            except Exception as e:  # noqa: BLE001
                ok = 0.0
                report["halo_error"] = repr(e)[:200]
            flag = torch.tensor([ok], dtype=torch.float64, device=device)
            self.comm.all_reduce_sum(flag, grp)
            if float(flag.item()) < len(grp):
                self.p2p = False
                report["halo"] = "all_gather (the batched p2p path failed its self-check)"
        # --- two networks in lockstep on two streams: one exchange group (partials, halo, tokens) of lane 0 on a side stream
        # interleaved with lane 1's on the caller's stream, in the order run_lockstep issues them; wrong data or an exception
        # switches the overlap off on every rank of the group (a hang cannot be caught here: see above)
        if self.two_streams and getattr(self.comm, "two_lanes", True) is False:
            self.two_streams = False                       # set by hand on a one-lane communicator: not a supported combination
            report["order_note"] = "two_streams requested on a one-lane communicator: switched off"
        report["order"] = "two streams, lockstep" if self.two_streams else "one stream"
        if self.two_streams:
            ok = 1.0
            try:
                cuda = device.type == "cuda"
                side = torch.cuda.Stream(device=device) if cuda else None
                import contextlib
                results = {}
                for step in ("partials", "halo", "tokens"):
                    for lane in (0, 1):
                        ctx = torch.cuda.stream(side) if (cuda and lane == 0) else contextlib.nullcontext()
                        self.lane = lane
                        with ctx:
                            if step == "partials":
                                buf, own = self.part_buffer(2, device)
                                own.copy_((pattern(lay.shard, lay.T_loc * 2)[:, :1] + 256.0 * lane).float().repeat(1, 64))
                                results[(step, lane)] = self.gather_partials(buf, 2)
                            elif step == "halo":
                                x = pattern(lay.shard, lay.T_loc * rows) + 512.0 * lane
                                results[(step, lane)] = self.halo_begin(x, rows)
                            else:
                                buf, own = self.kv_buffer(rows, C, device)
                                buf.zero_()
                                own.copy_(mine + 256.0 * lane)
                                results[(step, lane)] = (buf, self.kv_gather_begin(buf, rows))
                for lane in (0, 1):
                    ctx = torch.cuda.stream(side) if (cuda and lane == 0) else contextlib.nullcontext()
                    with ctx:
                        pb = results[("partials", lane)]
                        fp, fn = results[("halo", lane)].wait()
                        kb, work = results[("tokens", lane)]
                        work.wait()
                        for s_, (a, b) in enumerate(lay.bounds):
                            want = (pattern(s_, (b - a) * 2)[:, :1] + 256.0 * lane).float().repeat(1, 64)
                            if not torch.equal(pb[s_ * lay.T_max * 2:s_ * lay.T_max * 2 + (b - a) * 2], want):
                                ok = 0.0
                            n = (b - a) * rows
                            if not torch.equal(kb[s_ * lay.T_max * rows:s_ * lay.T_max * rows + n], ref[s_][:n] + 256.0 * lane):
                                ok = 0.0
                        if lay.prev_rank is not None:
                            ta, tb = lay.bounds[lay.shard - 1]
                            if not torch.equal(fp, (pattern(lay.shard - 1, (tb - ta) * rows) + 512.0 * lane)[-rows:]):
                                ok = 0.0
                        if lay.next_rank is not None and not torch.equal(fn, (pattern(lay.shard + 1, rows) + 512.0 * lane)):
                            ok = 0.0
                if cuda:
                    torch.cuda.current_stream(device).wait_stream(side)
            except Exception as e:  # noqa: BLE001
                ok = 0.0
                report["order_error"] = repr(e)[:200]
            self.lane = 0
            flag = torch.tensor([ok], dtype=torch.float64, device=device)
            self.comm.all_reduce_sum(flag, grp)
            if float(flag.item()) < len(grp):
                self.two_streams = False
                report["order"] = "one stream (the two-lane exchange failed its self-check)"
        return report

    # temporal GroupNorm statistics ----------------------------------------------------------------------------
    def part_buffer(self, nparts, device):
        """-> (buf fp32 [frame_ranks * T_max * nparts, 64], own = this shard's [T_loc * nparts, 64] rows of it): the gather
        buffer of the GroupNorm partials of one clip (one entry of 32 x {sum, sum of squares} per frame and row chunk).
        Cached per (network lane, stream, nparts) and zero-filled once: ``mofa_gn_partial_f16`` rewrites the own rows every
        time, the padding rows of a shorter shard stay zero (they are gathered and summed like any entry)."""
        lay = self.lay
        key = (self.lane, torch.cuda.current_stream().cuda_stream if device.type == "cuda" else 0, nparts, str(device))
        buf = self._part_bufs.get(key)
        if buf is None:
            buf = torch.zeros((lay.frame_ranks * lay.T_max * nparts, 64), dtype=torch.float32, device=device)
            self._part_bufs[key] = buf
        r0 = lay.shard * lay.T_max * nparts
        return buf, buf[r0:r0 + self.T_loc * nparts]

    def gather_partials(self, buf, nparts):
        """all ranks' partials into ``buf`` (in place; complete in stream order on return)"""
        if self.log is not None:
            self.log.append((self.lane, "partials"))
        return self.comm.gather_small_into(buf, self.lay.T_max * nparts, self.lay.frame_group, lane=self.lane)

    # temporal conv halo -----------------------------------------------------------------------------------
    def halo_begin(self, x, HW):
        """x [T_loc*HW, C]: starts the exchange of its first / last frame with the neighbour shards.  ``wait()`` ->
        (frame before this shard, frame after it), None at the clip ends (= the convolution's zero padding)."""
        lay = self.lay
        first, last = x[:HW], x[(self.T_loc - 1) * HW:]
        if self.log is not None:
            self.log.append((self.lane, "halo"))
        if self.p2p:
            return self.comm.halo_begin(first, last, lay.prev_rank, lay.next_rank, lay.frame_group, lane=self.lane)
        got = self.comm.all_gather(torch.cat([first, last], 0), lay.frame_group)      # conservative path (self_check)
        fp = got[lay.shard - 1][HW:] if lay.prev_rank is not None else None
        fn = got[lay.shard + 1][:HW] if lay.next_rank is not None else None
        return _HaloWork([], fp, fn)

    # temporal attention K/V ---------------------------------------------------------------------------------
    @property
    def kv_slots(self):
        """key-frame slots of the gathered K|V buffer (frame_ranks x largest shard); the masked attention entry takes
        at most 32"""
        return self.lay.frame_ranks * self.lay.T_max

    @property
    def kv_mask(self):
        """bit (shard * T_max + t) set for the frames shard holds: the padding frames of the shorter shards are clear"""
        m = 0
        for s_, (a, b) in enumerate(self.lay.bounds):
            m |= ((1 << (b - a)) - 1) << (s_ * self.lay.T_max)
        return m

    def kv_buffer(self, rows_per_frame, C2, device):
        """-> (buf [kv_slots * rows, C2] uninitialised, own [T_loc * rows, C2] = this rank's slot of it)"""
        lay = self.lay
        slot = lay.T_max * rows_per_frame
        buf = torch.empty((lay.frame_ranks * slot, C2), dtype=torch.float16, device=device)
        return buf, buf[lay.shard * slot:lay.shard * slot + self.T_loc * rows_per_frame]

    def kv_gather_begin(self, buf, rows_per_frame):
        if self.log is not None:
            self.log.append((self.lane, "tokens"))
        return self.comm.all_gather_into(buf, self.lay.T_max * rows_per_frame, self.lay.frame_group, lane=self.lane)

    def gather_frames(self, t, rows_per_frame):
        """t [T_loc*rows, C] (this shard's frames) -> [T_full*rows, C] in frame order (uneven shards are padded to
        the largest for the collective and compacted afterwards)."""
        lay = self.lay
        C = t.shape[1]
        pad_rows = lay.T_max * rows_per_frame
        if t.shape[0] != pad_rows:
            buf = torch.zeros((pad_rows, C), dtype=t.dtype, device=t.device)
            buf[:t.shape[0]].copy_(t)
        else:
            buf = t.contiguous()
        got = self.comm.all_gather(buf, lay.frame_group)
        out = torch.empty((self.T_full * rows_per_frame, C), dtype=t.dtype, device=t.device)
        for (a, b), g in zip(lay.bounds, got):
            out[a * rows_per_frame:b * rows_per_frame].copy_(g[:(b - a) * rows_per_frame])
        return out

    # CFG pair ---------------------------------------------------------------------------------------------
    def gather_cfg(self, noise):
        """noise [T_loc*HW, 4] of this half -> [2*T_loc*HW, 4] (unconditional first)"""
        got = self.comm.all_gather(noise, self.lay.pair_group)
        return torch.cat(got, 0) if len(got) > 1 else got[0]


# ---------------------------------------------------------------------------------------------------------
class GroupedWindowParallel:
    """Long video on more ranks than one window per rank can use: G = world / g groups of g ranks.  The distinct windows of a
    step are dealt round-robin to the GROUPS (as WindowParallel deals them to ranks) and every group runs its window
    frame-parallel (2-way CFG x g/2 frame shards inside the group, ``self.frame``); after a round ONE all-gather over all ranks
    hands every rank every stepped window (the g ranks of a group hold identical copies; entry 0 of each group is used).
    ``window_layout_costs`` says when this beats the two plain layouts (e.g. 4 windows on 8 ranks: 4 x 2)."""

    def __init__(self, comm, rank, world, ranks_per_group, window_size):
        g = ranks_per_group
        assert world % g == 0 and g >= 2 and g % 2 == 0
        self.comm, self.rank, self.world, self.g = comm, rank, world, g
        self.groups, self.slot = world // g, rank // g
        self.frame = FrameParallel(Layout(g, rank % g, window_size, base=self.slot * g), comm)

    @staticmethod
    def layout_of_rank(world, ranks_per_group, window_size):
        """the ``layout_of_rank`` argument of TorchComm for this arrangement"""
        g = ranks_per_group
        return lambda r: Layout(g, r % g, window_size, base=(r // g) * g)

    def rounds(self, keys):
        out = []
        for j in range(0, len(keys), self.groups):
            chunk = list(keys[j:j + self.groups])
            out.append(chunk + [None] * (self.groups - len(chunk)))
        return out

    def gather(self, t):
        return self.comm.all_gather_world(t)[::self.g]


class WindowParallel:
    """Long-video (Keypoint window loop) sharding: the DISTINCT temporal windows of one denoise step are independent
    (MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:470-511), so they are dealt round-robin to the ranks;
    after every round one all-gather hands every rank every stepped window, and all ranks apply the same overlap
    averaging.  No other exchange: each window runs the whole adapter + UNet on its own 25 frames."""

    def __init__(self, comm, rank, world):
        self.comm, self.rank, self.world = comm, rank, world
        self.slot = rank                     # position in a round (GroupedWindowParallel: the group)

    def rounds(self, keys):
        """keys: the distinct windows in view order -> list of rounds, each a list of ``world`` keys (None = idle slot)"""
        out = []
        for j in range(0, len(keys), self.world):
            chunk = list(keys[j:j + self.world])
            out.append(chunk + [None] * (self.world - len(chunk)))
        return out

    def gather(self, t):
        return self.comm.all_gather_world(t)


# ---------------------------------------------------------------------------------------------------------
# Which layout for a long video (Keypoint window loop) on R ranks?  A pure cost table, no device work.
# ---------------------------------------------------------------------------------------------------------
# time of ONE window step on g ranks (2-way CFG x g/2 frame shards) relative to one rank: device time per denoise step of rank 0
# of g on the 1-GPU proxy (tools/shard_proxy.py, loopback transport: no wire time, no waiting for peers -- LOWER bounds of the
# real times) over 236.3 ms for the whole clip on one GPU: 125.0 / 74.0 / 48.3 ms (profiles/r04_shard_proxy.log).  No multi-GPU
# node has measured them (every SCALE record of this build is a skipped one); callers with measured times pass ``step_time``.
WINDOW_STEP_TIME = {1: 1.0, 2: 0.53, 4: 0.31, 8: 0.20}


def window_layout_costs(windows, ranks, step_time=None):
    """-> list of (cost, name, groups, ranks_per_group) sorted by cost: the time of one denoise step over ``windows`` distinct
    windows on ``ranks`` ranks, in units of one window step on one rank, for every way this package can spread them:

      "window"   WindowParallel: one window per rank and round, ceil(windows / ranks) rounds (ranks beyond the windows idle);
      "frame"    FrameParallel with a Layout of window_size frames: every window on ALL ranks, one after the other;
      "groups"   G groups of g = ranks / G ranks (g even): the windows dealt to the groups round-robin, every group runs its
                 window frame-parallel -- WindowParallel over group leaders composed with FrameParallel inside a group.

    With 7 windows on 8 ranks no split beats "window" (cost 1.0, one rank idle): a window finishes sooner only on >= 2 ranks,
    seven windows on >= 2 ranks each need 14 rank slots, i.e. two rounds of >= 0.53 each.  The eighth rank's share of the work is
    the VAE decode (13 chunks dealt over all 8 ranks after the loop)."""
    st = dict(WINDOW_STEP_TIME)
    st.update(step_time or {})
    out = []
    out.append((float(-(-windows // ranks)) * st[1], "window", ranks, 1))
    if ranks in st and ranks >= 2:
        out.append((windows * st[ranks], "frame", 1, ranks))
    g = 2
    while g < ranks:
        if ranks % g == 0 and g in st:
            G = ranks // g
            out.append((float(-(-windows // G)) * st[g], "groups", G, g))
        g *= 2
    return sorted(out, key=lambda c: (c[0], {"window": 0, "frame": 1, "groups": 2}[c[1]]))


def plan_windows(windows, ranks, step_time=None):
    """the cheapest layout of ``window_layout_costs`` as (name, groups, ranks_per_group)"""
    c = window_layout_costs(windows, ranks, step_time)[0]
    return c[1], c[2], c[3]
