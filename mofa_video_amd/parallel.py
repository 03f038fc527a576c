"""Clip partitioning across the GPUs of one MI355X node -- new work, the reference is single-GPU.

Layout for N ranks (N in 1, 2, 4, 8, ...):  2-way CFG x (N/2)-way frames.
    rank r:  half = r // frame_ranks  (0 = unconditional, 1 = conditional)   [N == 1: both halves on the one rank]
             shard = r %  frame_ranks -> frames [f0, f1) of the clip (contiguous, sizes differ by at most 1)
Weights are replicated.  Exchanges inside one denoise step (RCCL over xGMI through torch.distributed):
    * temporal GroupNorm (statistics span all T frames): all-reduce of fp64 [32][2] partial sums   (frame group)
    * temporal (3,1,1) convolution: one halo frame from each neighbour shard (batched p2p)        (frame group)
    * temporal self-attention: ONE all_gather_into_tensor of the normed hidden tokens (C columns; r02 gathered K|V = 2C) into
      a preallocated [frame_ranks x T_max frames] buffer (the LayerNorm writes this rank's slot in place, the collective runs
      asynchronously under the Q projection, every rank then projects K|V for all key slots, and the attention kernel masks
      the padding frames of uneven shards -- no pad / concat / compaction copies)                   (frame group)
    * CFG combine: the two halves of one frame shard swap their noise predictions                 (pair group)
and once per clip: all-gather of the final latents before the VAE decode, whose chunks are independent and are
dealt round-robin to ALL ranks.  Everything per-frame (2-D convs, spatial norms/attention, FFs, the adapter warps,
zero convs, the Euler step) needs no communication.

``Comm`` implementations: ``TorchComm`` (torch.distributed: "nccl" = RCCL on the GPUs, "gloo" in the CPU tests) and
``ThreadComm`` (virtual ranks as threads of one process -- lets the whole sharded HIP path be checked against the
unsharded one on a single GPU).
"""
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
    def __init__(self, world, rank, T, cfg_ranks=None):
        """cfg_ranks: None = 2 whenever world >= 2 (the product layout).  1 = frames only (both CFG halves on every
        rank) -- used by the exchange-primitive tests to shard frames over 2 ranks; the pipelines use the default."""
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
        self.frame_group = [h * self.frame_ranks + s for s in range(self.frame_ranks)]
        self.pair_group = [self.shard, self.frame_ranks + self.shard] if self.cfg_ranks == 2 else [rank]
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

    def __init__(self, layout_of_rank):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        world = dist.get_world_size()
        self._groups = {}
        seen = []
        for r in range(world):
            lay = layout_of_rank(r)
            for g in (tuple(lay.frame_group), tuple(lay.pair_group)):
                if g not in seen:
                    seen.append(g)
        for g in seen:
            self._groups[g] = dist.new_group(list(g)) if len(g) > 1 else None
        self._world_group = None

    def _g(self, ranks):
        return self._groups[tuple(ranks)]

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

    def all_gather_into(self, buf, slot_rows, ranks):
        """In-place all-gather: ``buf`` [len(ranks) * slot_rows, C] already holds this rank's rows in its own slot; after
        ``wait()`` it holds every rank's.  Asynchronous: RCCL runs the collective on its own stream, kernels launched
        between this call and ``wait()`` overlap it (``wait`` makes the CURRENT stream wait, the host does not block)."""
        if len(ranks) == 1:
            return _Done()
        i = list(ranks).index(self.rank)
        return self.dist.all_gather_into_tensor(buf, buf[i * slot_rows:(i + 1) * slot_rows], group=self._g(ranks),
                                                async_op=True)

    def exchange_halo(self, first, last, prev_rank, next_rank):
        """send `first` to prev and `last` to next; receive prev's last and next's first (None at the clip ends)"""
        dist = self.dist
        ops, from_prev, from_next = [], None, None
        if prev_rank is not None:
            from_prev = torch.empty_like(last)
            ops += [dist.P2POp(dist.isend, first.contiguous(), prev_rank), dist.P2POp(dist.irecv, from_prev, prev_rank)]
        if next_rank is not None:
            from_next = torch.empty_like(first)
            ops += [dist.P2POp(dist.isend, last.contiguous(), next_rank), dist.P2POp(dist.irecv, from_next, next_rank)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return from_prev, from_next


class _Done:
    def wait(self):
        return True


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
    def __init__(self, tworld, rank):
        self.tw, self.rank = tworld, rank

    def _exchange(self, t, ranks):
        bar, slots = self.tw._group_state(ranks)
        slots[list(ranks).index(self.rank)] = t
        bar.wait()
        got = list(slots)
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
        return [g.clone() for g in self._exchange(t.contiguous(), ranks)]

    def all_gather_world(self, t):
        return self.all_gather(t, list(range(self.tw.world)))

    def all_gather_into(self, buf, slot_rows, ranks):
        if len(ranks) == 1:
            return _Done()
        i = list(ranks).index(self.rank)
        got = self._exchange(buf[i * slot_rows:(i + 1) * slot_rows], ranks)
        for j, g in enumerate(got):
            if j != i:
                buf[j * slot_rows:(j + 1) * slot_rows].copy_(g)
        if buf.is_cuda:
            torch.cuda.current_stream().synchronize()      # the peers' slots must stay put until they were read
        bar, _ = self.tw._group_state(ranks)
        bar.wait()
        return _Done()


# ---------------------------------------------------------------------------------------------------------
class FrameParallel:
    """The per-rank object blocks consult (``Ctx.par``) when a clip's frames are sharded."""

    def __init__(self, layout, comm, p2p=True):
        self.lay, self.comm = layout, comm
        self.T_full, self.T_loc, self.f0, self.f1 = layout.T, layout.T_loc, layout.f0, layout.f1
        self.p2p = p2p and isinstance(comm, TorchComm)
        self.kv_inplace = True      # temporal attention K|V: in-place asynchronous all_gather_into_tensor (else: compacting gather)
        self.gather_hidden = True   # ... of the normed hidden tokens (C columns; K|V projected after the gather) instead of K|V (2C)

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
                fp, fn = self.comm.exchange_halo(first, last, lay.prev_rank, lay.next_rank)
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
        return report

    # temporal GroupNorm -----------------------------------------------------------------------------------
    def reduce_gn(self, sums):
        return self.comm.all_reduce_sum(sums, self.lay.frame_group)

    # temporal conv halo -----------------------------------------------------------------------------------
    def halo(self, x, HW):
        """x [T_loc*HW, C] -> [(T_loc+2)*HW, C]: neighbour shards' boundary frames before/after, zeros at clip ends
        (= the conv's zero padding)."""
        lay = self.lay
        C = x.shape[1]
        ext = torch.empty(((self.T_loc + 2) * HW, C), dtype=x.dtype, device=x.device)
        ext[HW:(self.T_loc + 1) * HW].copy_(x)
        first, last = x[:HW], x[(self.T_loc - 1) * HW:]
        if self.p2p:
            fp, fn = self.comm.exchange_halo(first, last, lay.prev_rank, lay.next_rank)
        else:
            both = torch.cat([first, last], 0)
            got = self.comm.all_gather(both, lay.frame_group)
            fp = got[lay.shard - 1][HW:] if lay.prev_rank is not None else None
            fn = got[lay.shard + 1][:HW] if lay.next_rank is not None else None
        if fp is not None:
            ext[:HW].copy_(fp)
        else:
            ext[:HW].zero_()
        if fn is not None:
            ext[(self.T_loc + 1) * HW:].copy_(fn)
        else:
            ext[(self.T_loc + 1) * HW:].zero_()
        return ext

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
        return self.comm.all_gather_into(buf, self.lay.T_max * rows_per_frame, self.lay.frame_group)

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
class WindowParallel:
    """Long-video (Keypoint window loop) sharding: the DISTINCT temporal windows of one denoise step are independent
    (MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:470-511), so they are dealt round-robin to the ranks;
    after every round one all-gather hands every rank every stepped window, and all ranks apply the same overlap
    averaging.  No other exchange: each window runs the whole adapter + UNet on its own 25 frames."""

    def __init__(self, comm, rank, world):
        self.comm, self.rank, self.world = comm, rank, world

    def rounds(self, keys):
        """keys: the distinct windows in view order -> list of rounds, each a list of ``world`` keys (None = idle slot)"""
        out = []
        for j in range(0, len(keys), self.world):
            chunk = list(keys[j:j + self.world])
            out.append(chunk + [None] * (self.world - len(chunk)))
        return out

    def gather(self, t):
        return self.comm.all_gather_world(t)
