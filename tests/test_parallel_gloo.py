"""CPU tests (gloo, world_size 2 and 4, 127.0.0.1) of the clip-partitioning layer ``mofa_video_amd/parallel.py``:
the layout arithmetic and every exchange primitive of the frame-sharded path, each checked against the
single-process result computed with plain torch on the full clip."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from mofa_video_amd.parallel import FrameParallel, Layout, TorchComm, split_frames


def test_layout_arithmetic():
    assert split_frames(25, 4) == [(0, 7), (7, 13), (13, 19), (19, 25)]
    assert split_frames(25, 1) == [(0, 25)]
    l = Layout(8, 5, 25)            # 2-way CFG x 4-way frames
    assert (l.cfg_ranks, l.frame_ranks, l.half, l.shard) == (2, 4, 1, 1)
    assert (l.f0, l.f1, l.T_loc, l.T_max) == (7, 13, 6, 7)
    assert l.frame_group == [4, 5, 6, 7] and l.pair_group == [1, 5]
    assert (l.prev_rank, l.next_rank) == (4, 6)
    l = Layout(2, 1, 25)
    assert (l.half, l.frame_group, l.pair_group, l.sharded_frames, l.B_loc) == (1, [1], [0, 1], False, 1)
    l = Layout(1, 0, 25)
    assert (l.half, l.B_loc, l.T_loc, l.pair_group) == (None, 2, 25, [0])
    # every frame is owned exactly once per CFG half
    for world in (2, 4, 8):
        owned = {}
        for r in range(world):
            lay = Layout(world, r, 25)
            for f in range(lay.f0, lay.f1):
                owned.setdefault((lay.half, f), []).append(r)
        assert len(owned) == 2 * 25 and all(len(v) == 1 for v in owned.values())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, HW, C):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = Layout(world, rank, T)
        comm = TorchComm(lambda r: Layout(world, r, T))
        par = FrameParallel(lay, comm)
        g = torch.Generator().manual_seed(1234)              # identical "full clip" data on every rank
        full = torch.randn(2, T * HW, C, generator=g)          # [half, frames*HW, C]
        h = lay.half if lay.half is not None else 0
        mine = full[h, lay.f0 * HW:lay.f1 * HW].contiguous()

        # 1. halo exchange (asynchronous begin / wait) reproduces the zero-padded (3,1,1) convolution of the whole clip
        w = torch.randn(C, C, 3, 1, 1, generator=g)
        x5 = full[h].reshape(1, T, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)          # [1,C,T,HW,1]
        ref = F.conv3d(x5, w, padding=(1, 0, 0))[0, :, lay.f0:lay.f1, :, 0]             # [C, T_loc, HW]
        fp, fn = par.halo_begin(mine, HW).wait()
        assert (fp is None) == (lay.f0 == 0) and (fn is None) == (lay.f1 == T)
        z = torch.zeros(HW, C)
        ext = torch.cat([z if fp is None else fp, mine, z if fn is None else fn], 0)
        e5 = ext.reshape(1, lay.T_loc + 2, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)
        got = F.conv3d(e5, w)[0, :, :, :, 0]                                             # valid conv over the halo'd shard
        assert torch.allclose(got, ref, atol=1e-4), (rank, (got - ref).abs().max())

        # 2. temporal GroupNorm statistics: the gathered partials (zero rows for the padding frames of shorter shards) sum to the
        #    sums over the whole clip, identically on every rank; the buffer is cached and its padding rows stay zero
        for rep in range(2):
            buf, own = par.part_buffer(3, mine.device)
            assert own.shape == (lay.T_loc * 3, 64) and buf.shape == (lay.frame_ranks * lay.T_max * 3, 64)
            fr = mine.reshape(lay.T_loc, HW, C).double()
            for t in range(lay.T_loc):                                                   # 3 "row chunks" of 2 rows per frame
                for ch in range(3):
                    blk = fr[t, ch * 2:(ch + 1) * 2, :32]
                    own[t * 3 + ch] = torch.stack([blk.sum(0), (blk ** 2).sum(0)], -1).reshape(64).float()
            par.gather_partials(buf, 3)
            tot = buf.double().reshape(-1, 32, 2).sum(0)
            fs = torch.stack([full[h][:, :32].double().sum(0), (full[h][:, :32].double() ** 2).sum(0)], -1)
            assert torch.allclose(tot, fs, rtol=1e-5, atol=1e-3), (rank, rep, (tot - fs).abs().max())
            assert par.part_buffer(3, mine.device)[0] is buf

        # 3. K|V all-gather along the frame axis (uneven shards are padded and compacted)
        kv = par.gather_frames(mine, HW)
        assert torch.equal(kv, full[h])
        lat = par.gather_frames(mine.reshape(lay.T_loc, HW * C), 1)
        assert torch.equal(lat, full[h].reshape(T, HW * C))

        # 3b. the in-place K|V all-gather of the temporal attention: this shard's rows are written into its own slot of a
        #     [frame_ranks x T_max frames] buffer, ONE all_gather_into_tensor fills the rest, the padding frames of the
        #     shorter shards stay untouched and are excluded by the key mask
        buf, own = par.kv_buffer(HW, C, mine.device)
        assert own.shape == mine.shape and par.kv_slots == lay.frame_ranks * lay.T_max
        buf.fill_(float("nan"))
        own.copy_(mine.half())
        par.kv_gather_begin(buf, HW).wait()
        mask = par.kv_mask
        assert bin(mask).count("1") == T
        for s_, (a, b) in enumerate(lay.bounds):
            for t in range(lay.T_max):
                slot = buf[(s_ * lay.T_max + t) * HW:(s_ * lay.T_max + t + 1) * HW]
                if t < b - a:
                    assert (mask >> (s_ * lay.T_max + t)) & 1
                    assert torch.equal(slot, full[h, (a + t) * HW:(a + t + 1) * HW].half()), (rank, s_, t)
                else:
                    assert not (mask >> (s_ * lay.T_max + t)) & 1
                    assert torch.isnan(slot).all()                      # padding frames: never written

        # 4. CFG pair exchange: unconditional half first
        if world >= 2:
            both = par.gather_cfg(mine[:, :4].contiguous())
            exp = torch.cat([full[0, lay.f0 * HW:lay.f1 * HW, :4], full[1, lay.f0 * HW:lay.f1 * HW, :4]], 0)
            assert torch.equal(both, exp)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 5), (4, 5), (4, 7)])
def test_frame_parallel_exchanges_gloo(world, T):
    mp.spawn(_worker, args=(world, _free_port(), T, 6, 32), nprocs=world, join=True)


def _window_worker(rank, world, port, nwin):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mofa_video_amd.parallel import WindowParallel
        wp = WindowParallel(TorchComm(lambda r: Layout(world, r, 25)), rank, world)
        keys = [(1 + 3 * i, 4 + 3 * i) for i in range(nwin)]
        rounds = wp.rounds(keys)
        assert [k for rnd in rounds for k in rnd if k is not None] == keys and all(len(r) == world for r in rounds)
        done = {}
        for rnd in rounds:
            mine = rnd[rank]
            t = torch.full((4, 3), float(mine[0]) if mine is not None else 0.0)
            for key, got in zip(rnd, wp.gather(t)):
                if key is not None:
                    done[key] = got
        assert sorted(done) == keys
        for k, v in done.items():                 # every rank holds every window, each produced by its owner
            assert torch.equal(v, torch.full((4, 3), float(k[0])))
    finally:
        dist.destroy_process_group()


def _frames_only_worker(rank, world, port, T, HW, C):
    """cfg_ranks = 1: the frames of BOTH halves sharded over 2 ranks (the layout the 2-GPU nccl test also uses)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frame_exchange_checks import check_frame_exchanges
        check_frame_exchanges(rank, world, T, HW, C, "cpu")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T", [4, 5])
def test_frames_only_layout_gloo(T):
    mp.spawn(_frames_only_worker, args=(2, _free_port(), T, 6, 32), nprocs=2, join=True)


@pytest.mark.parametrize("world,nwin", [(2, 3), (2, 4), (4, 3)])
def test_window_parallel_gloo(world, nwin):
    mp.spawn(_window_worker, args=(world, _free_port(), nwin), nprocs=world, join=True)


def test_deal_ready_chunks_prefers_idle_ranks():
    """the common table that says which rank decodes a VAE chunk that became final before the last round of the last step"""
    from mofa_video_amd.pipeline import _deal_ready_chunks
    assert _deal_ready_chunks([0, 1, 2, 3, 4, 5], 4, busy_next=[0, 1, 2]) == [(0, 3), (1, 3), (2, 3), (3, 0), (4, 1), (5, 2)]
    assert _deal_ready_chunks([0, 1, 2], 1, busy_next=[0]) == [(0, 0), (1, 0)]      # one GPU: two per round, the rest wait
    assert _deal_ready_chunks([], 8, busy_next=list(range(7))) == []


# ---- the sharded temporal block (GroupNorm over the clip + SiLU -> (3,1,1) convolution) end to end on CPU ---------------------
class _Norm:
    def __init__(self, C, g):
        self.g, self.b, self.eps = torch.randn(C, generator=g) * 0.3 + 1.0, torch.randn(C, generator=g) * 0.2, 1e-5


class _Conv:
    def __init__(self, C, g):
        w = torch.randn(C, C, 3, 1, 1, generator=g) * (3 * C) ** -0.5
        self.w5 = w.half().float()
        self.w = w[:, :, :, 0, 0].permute(0, 2, 1).reshape(C, 3 * C).half().contiguous()    # [N][tap][Cin] (weights.pack_conv3d_t3)
        self.b = torch.randn(C, generator=g) * 0.1


def _reference_block(full, norm, conv, T, HW, C, res=None, s_acc=1.0):
    """GroupNorm(32) over all T*HW positions of the clip + SiLU -> Conv3d (3,1,1), zero padding, (+ residual), plain torch"""
    x = full.float().reshape(1, T, HW, C).permute(0, 3, 1, 2)                                # [1, C, T, HW]
    y = F.silu(F.group_norm(x, 32, norm.g, norm.b, norm.eps)).half().float()
    o = F.conv3d(y.unsqueeze(-1), conv.w5, conv.b, padding=(1, 0, 0))[0, :, :, :, 0]         # [C, T, HW]
    o = o.permute(1, 2, 0).reshape(T * HW, C) * s_acc
    return o + res.float() if res is not None else o


def _block_worker(rank, world, port, T, HW, C, two_networks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        import time
        import emu_ops
        from mofa_video_amd import blocks
        emu_ops.install()
        lay = Layout(world, rank, T, cfg_ranks=1)
        par = FrameParallel(lay, TorchComm(lambda r: Layout(world, r, T, cfg_ranks=1)))
        c = blocks.Ctx(1, lay.T_loc)
        c.par = par

        def network(seed, nblocks, errs, delay=0.0):
            """layer generator: ``nblocks`` sharded norm + conv(3,1,1) blocks on seeded data, checked against the whole clip"""
            g = torch.Generator().manual_seed(seed)
            for k in range(nblocks):
                full = (torch.randn(T * HW, C, generator=g) * 1.5 + 0.3).half()
                res = torch.randn(T * HW, C, generator=g).half()
                n1, c1 = _Norm(C, g), _Conv(C, g)
                time.sleep(delay * ((k + rank) % 3))                                       # perturb the host timing per rank
                mine, rmine = full[lay.f0 * HW:lay.f1 * HW].contiguous(), res[lay.f0 * HW:lay.f1 * HW].contiguous()
                got = blocks._sharded_norm_convt3(n1, c1, mine, c, HW, r1=rmine, s1=1.0, s_acc=0.7)
                ref = _reference_block(full, n1, c1, T, HW, C, res, 0.7)[lay.f0 * HW:lay.f1 * HW]
                errs.append(float((got.float() - ref).abs().max()))
                yield
            return errs
        if not two_networks:
            for split in (True, False):
                par.split_convs = split
                errs = blocks.drive(network(7, 3, []))
                assert max(errs) < 2e-2, (rank, split, errs)
        else:
            # two networks of different length enqueued in lockstep by ONE thread, as pipeline._denoise_forward_sharded does:
            # every rank issues the same sequence of exchanges whatever its host timing is (the ranks are delayed differently)
            par.log = []

            def lane(i):
                @contextlib.contextmanager
                def cm():
                    par.lane = i
                    yield
                return cm
            e0, e1 = blocks.run_lockstep([network(11, 4, [], 0.02 if rank == 0 else 0.0), network(12, 6, [], 0.02 if rank == 1 else 0.0)],
                                         [lane(0), lane(1)])
            assert max(e0) < 2e-2 and max(e1) < 2e-2 and len(e0) == 4 and len(e1) == 6, (rank, e0, e1)
            want = [(ln, kind) for k in range(4) for ln in (0, 1) for kind in ("halo", "partials")] + [(1, "halo"), (1, "partials")] * 2
            assert par.log == want, (rank, par.log)
            logs = [None] * world
            dist.all_gather_object(logs, par.log)
            assert all(lg == logs[0] for lg in logs)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 7), (4, 17), (3, 8)])
def test_sharded_temporal_block_gloo(world, T):
    """shards of 4+3 / 5+4+4+4 / 3+3+2 frames: interior + boundary launches where a shard has >= 4 frames, one launch otherwise"""
    mp.spawn(_block_worker, args=(world, _free_port(), T, 6, 32, False), nprocs=world, join=True)


def test_two_networks_in_lockstep_gloo():
    mp.spawn(_block_worker, args=(2, _free_port(), 9, 6, 32, True), nprocs=2, join=True)


def test_run_lockstep_order_and_values():
    """blocks.run_lockstep: one layer of each live generator in turn, each inside its own context; values returned in order"""
    import contextlib
    from mofa_video_amd.blocks import drive, run_lockstep
    trace = []

    def gen(name, n):
        for k in range(n):
            trace.append((name, k, ctx[0]))
            yield
        return name * n

    ctx = [None]

    def enter(tag):
        @contextlib.contextmanager
        def cm():
            ctx[0] = tag
            yield
            ctx[0] = None
        return cm
    vals = run_lockstep([gen("a", 2), gen("b", 4)], [enter("A"), enter("B")])
    assert vals == ["aa", "bbbb"]
    assert trace == [("a", 0, "A"), ("b", 0, "B"), ("a", 1, "A"), ("b", 1, "B"), ("b", 2, "B"), ("b", 3, "B")]
    assert drive(gen("c", 3)) == "ccc"


def test_grouped_layout_and_window_cost_table():
    """GroupedWindowParallel's group layouts hold GLOBAL ranks; the cost table is a pure function of (windows, ranks)"""
    from mofa_video_amd.parallel import GroupedWindowParallel, plan_windows, window_layout_costs
    lay = Layout(4, 1, 25, base=4)                   # second group of 4 on 8 ranks: 2-way CFG x 2 frame shards on ranks 4..7
    assert (lay.half, lay.shard, lay.frame_group, lay.pair_group) == (0, 1, [4, 5], [5, 7])
    assert (lay.prev_rank, lay.next_rank) == (4, None)
    lor = GroupedWindowParallel.layout_of_rank(8, 4, 25)
    groups = {tuple(lor(r).frame_group) for r in range(8)}
    assert groups == {(0, 1), (2, 3), (4, 5), (6, 7)}
    assert {tuple(lor(r).pair_group) for r in range(8)} == {(0, 2), (1, 3), (4, 6), (5, 7)}
    # BASELINE config 5: 7 windows on 8 ranks -> one window per rank, one rank idle (any split needs two rounds)
    assert plan_windows(7, 8) == ("window", 8, 1)
    costs = {(n, G, g): c for c, n, G, g in window_layout_costs(7, 8)}
    assert costs[("window", 8, 1)] == 1.0 and costs[("groups", 4, 2)] == pytest.approx(2 * 0.53)
    assert plan_windows(4, 8) == ("groups", 4, 2)    # fewer windows than half the ranks: CFG pairs
    assert plan_windows(1, 8) == ("frame", 1, 8) and plan_windows(2, 8)[0] == "groups"
    assert plan_windows(15, 8) == ("window", 8, 1) and plan_windows(7, 4) == ("window", 4, 1)
    # measured step times replace the assumed ones: with perfect 8-way scaling frame sharding wins for 7 windows
    assert plan_windows(7, 8, {8: 0.125})[0] == "frame"


# ---------------------------------------------------------------------------------------------------------
# the WHOLE UNet forward of a rank of the 2-way CFG x frame-shard layout over gloo against the unsharded forward
# (host graph + exchange protocol end to end; the HIP entry points are the torch stand-ins of tests/emu_ops.py)
# ---------------------------------------------------------------------------------------------------------
def _unet_worker(rank, world, port, T, cfg_name="TINY"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import emu_ops
        import helpers
        from helpers import rel_l2
        TINY = getattr(helpers, cfg_name)      # LDMK_UNET: 320 channels at level 0 -> the fused feed-forward path (ops.ff320)
        emu_ops.install()
        from mofa_video_amd import ops, schema
        from mofa_video_amd.parallel import FrameParallel, Layout, TorchComm
        from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
        torch.set_num_threads(2)
        hu = UNetSpatioTemporalConditionControlNetModel(schema.synthetic_state_dict(schema.unet_schema(TINY), seed=3), config=TINY,
                                                        device="cpu")
        H = W = 256
        h, w = H // 8, W // 8
        g = torch.Generator().manual_seed(5)                       # the same clip on every rank
        x = ops.nchw_to_tokens(torch.randn(2 * T, 8, h, w, generator=g), ld=hu.in_ld)
        emb = torch.randn(2, 1, TINY["cross_attention_dim"], generator=g)
        ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
        boc = TINY["block_out_channels"]
        dims = [(boc[0], h * w)] * 3 + [(boc[0], h * w // 4)] + [(boc[1], h * w // 4)] * 2 + [(boc[1], h * w // 16)] + \
               [(boc[2], h * w // 16)] * 2 + [(boc[2], h * w // 64)] + [(boc[3], h * w // 64)] * 2
        down = [(torch.randn(2 * T * hw, Cc, generator=g) * 0.3).half() for Cc, hw in dims]
        mid = (torch.randn(2 * T * h * w // 64, boc[3], generator=g) * 0.3).half()
        ref = hu.forward_tokens(x, hu.make_ctx(0.7, emb, ids, 2, T), h, w, down, mid)          # [2 T h w, 4], unsharded
        lay = Layout(world, rank, T)
        par = FrameParallel(lay, TorchComm(lambda r: Layout(world, r, T)))

        def rows(t):
            hw = t.shape[0] // (2 * T)
            return t[(lay.half * T + lay.f0) * hw:(lay.half * T + lay.f1) * hw]
        c = hu.make_ctx(0.7, emb, ids, 1, lay.T_loc, half=lay.half, par=par if lay.sharded_frames else None)
        out = hu.forward_tokens(rows(x), c, h, w, [rows(d) for d in down], rows(mid))
        e = rel_l2(out, rows(ref))
        assert out.shape == rows(ref).shape and e < 2e-3, (rank, e)
        errs = [None] * world
        dist.all_gather_object(errs, e)
        if rank == 0:
            print("sharded UNet forward vs unsharded, rel-L2 per rank:", ["%.2e" % v for v in errs])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 4), (4, 5), (8, 9)])
def test_sharded_unet_forward_gloo(world, T):
    """2-way CFG x {1, 2, 4} frame shards (uneven: 5 = 3 + 2, 9 = 3 + 2 + 2 + 2): every rank's noise prediction for its frames against
    the rows of the unsharded forward.  Stated bound 2e-3 (measured 8e-4 at every world size, the CFG-only split included): the CPU
    stand-ins run torch's sgemm, whose blocking -- hence fp32 summation order -- depends on the row count, so a half-size launch
    flips last fp16 bits that then propagate; on the GPU the same comparison is bit-identical at world 2 and 8e-4 at 4 / 8
    (tests/test_sharded_gpu.py), because a tile's arithmetic does not depend on how many tiles the launch has."""
    mp.spawn(_unet_worker, args=(world, _free_port(), T), nprocs=world, join=True)


def test_sharded_unet_forward_fused_ff_gloo():
    """the same with 320 channels at level 0: the level-0 feed-forwards take the fused launch (blocks.GegluFF.fused), whose second
    output -- norm1 of the temporal block -- is written straight into the rank's slot of the hidden-token all-gather buffer"""
    mp.spawn(_unet_worker, args=(4, _free_port(), 4, "LDMK_UNET"), nprocs=4, join=True)
