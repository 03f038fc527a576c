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

        # 1. halo exchange reproduces the zero-padded (3,1,1) convolution of the whole clip
        w = torch.randn(C, C, 3, 1, 1, generator=g)
        x5 = full[h].reshape(1, T, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)          # [1,C,T,HW,1]
        ref = F.conv3d(x5, w, padding=(1, 0, 0))[0, :, lay.f0:lay.f1, :, 0]             # [C, T_loc, HW]
        ext = par.halo(mine, HW)
        assert ext.shape[0] == (lay.T_loc + 2) * HW
        e5 = ext.reshape(1, lay.T_loc + 2, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)
        got = F.conv3d(e5, w)[0, :, :, :, 0]                                             # valid conv over the halo'd shard
        assert torch.allclose(got, ref, atol=1e-4), (rank, (got - ref).abs().max())

        # 2. temporal GroupNorm statistics: all-reduced partial sums == sums over the whole clip
        sums = torch.stack([mine.double().sum(0), (mine.double() ** 2).sum(0)], -1)[:32].reshape(1, -1, 2).clone()
        red = par.reduce_gn(sums.clone())
        fs = torch.stack([full[h].double().sum(0), (full[h].double() ** 2).sum(0)], -1)[:32].reshape(1, -1, 2)
        assert torch.allclose(red, fs, rtol=1e-12)

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
