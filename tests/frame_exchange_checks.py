"""The frame-shard exchange checks shared by the gloo (CPU) and nccl = RCCL (2-GPU) tests: world ranks, cfg_ranks = 1, so the
frames of a clip are sharded over ALL ranks and every primitive of ``parallel.FrameParallel`` moves real data."""
import torch


def check_frame_exchanges(rank, world, T, HW, C, device):
    from mofa_video_amd.parallel import FrameParallel, Layout, TorchComm
    lay = Layout(world, rank, T, cfg_ranks=1)
    assert lay.frame_ranks == world and lay.half is None and lay.B_loc == 2
    comm = TorchComm(lambda r: Layout(world, r, T, cfg_ranks=1))
    par = FrameParallel(lay, comm)
    g = torch.Generator().manual_seed(99)
    full = torch.randn(T * HW, C, generator=g).half().to(device)
    mine = full[lay.f0 * HW:lay.f1 * HW].contiguous()

    fp, fn = par.halo_begin(mine, HW).wait()                              # neighbour frames / None at the clip ends
    if str(device) != "cpu":
        torch.cuda.synchronize()
    assert (fp is None) == (lay.f0 == 0) and (fn is None) == (lay.f1 == T)
    if fp is not None:
        assert torch.equal(fp, full[(lay.f0 - 1) * HW:lay.f0 * HW])
    if fn is not None:
        assert torch.equal(fn, full[lay.f1 * HW:(lay.f1 + 1) * HW])

    pbuf, pown = par.part_buffer(2, torch.device(device))                 # GroupNorm partials: gathered, padding rows zero
    pown.copy_(torch.arange(lay.T_loc * 2 * 64, dtype=torch.float32, device=device).reshape(-1, 64) + 1000.0 * lay.shard)
    par.gather_partials(pbuf, 2)
    if str(device) != "cpu":
        torch.cuda.synchronize()
    for s_, (a, b) in enumerate(lay.bounds):
        rows = pbuf[s_ * lay.T_max * 2:(s_ * lay.T_max + b - a) * 2]
        exp = torch.arange((b - a) * 2 * 64, dtype=torch.float32, device=device).reshape(-1, 64) + 1000.0 * s_
        assert torch.equal(rows, exp), (rank, s_)
        assert not pbuf[(s_ * lay.T_max + b - a) * 2:(s_ + 1) * lay.T_max * 2].any()

    buf, own = par.kv_buffer(HW, C, device)                               # in-place all_gather_into_tensor + key mask
    buf.fill_(float("nan"))
    own.copy_(mine)
    work = par.kv_gather_begin(buf, HW)
    work.wait()
    if str(device) != "cpu":
        torch.cuda.synchronize()
    for s_, (a, b) in enumerate(lay.bounds):
        rows = buf[s_ * lay.T_max * HW:(s_ * lay.T_max + b - a) * HW]
        assert torch.equal(rows, full[a * HW:b * HW]), (rank, s_)
    assert bin(par.kv_mask).count("1") == T

    assert torch.equal(par.gather_frames(mine, HW), full)                 # the compacting gather (final latents)

    # transport self-check: both fast paths pass on a working transport; a path that delivers wrong data is switched off on
    # EVERY rank of the group (here: rank 0's in-place gather is sabotaged) and the conservative path still works
    rep = par.self_check(device)
    assert par.kv_inplace and rep["kv_gather"].startswith("in-place"), rep
    orig = par.kv_gather_begin
    if rank == 0:
        def broken(buf_, rows_):
            w = orig(buf_, rows_)
            w.wait()
            buf_[lay.T_max * rows_, 0] = 777.0                             # corrupt the first element of shard 1's slot
            return w
        par.kv_gather_begin = broken
    rep = par.self_check(device)
    par.kv_gather_begin = orig
    assert not par.kv_inplace and "compacting" in rep["kv_gather"], (rank, rep)
    assert torch.equal(par.gather_frames(mine, HW), full)
    par.kv_inplace = True
    # RCCL / gloo transports run the single-stream order unless opted in (round-4 advice); with two lanes every network has its
    # own bulk / ctl / data communicators and the self-check exercises one interleaved exchange group on both of them
    assert not par.two_streams and rep["order"] == "one stream", rep
    par2 = FrameParallel(lay, TorchComm(lambda r: Layout(world, r, T, cfg_ranks=1), two_lanes=True))
    assert par2.two_streams                                                # derived from the communicator's own two_lanes
    par.two_streams = True                                                 # by hand on the ONE-lane communicator: refused
    rep1 = par.self_check(device)
    assert not par.two_streams and "order_note" in rep1, (rank, rep1)
    rep2 = par2.self_check(device)
    assert par2.two_streams and rep2["order"].startswith("two streams") and "order_error" not in rep2, (rank, rep2)
    orig_h = par2.halo_begin
    if rank == world - 1:                                                  # lane 1's halo frames arrive corrupted on ONE rank
        def broken_h(x, HW_):
            w = orig_h(x, HW_)
            if par2.lane == 1 and w.res[0] is not None:
                w.wait()
                w.res[0][0, 0] += 1.0
            return w
        par2.halo_begin = broken_h
    rep2 = par2.self_check(device)
    par2.halo_begin = orig_h
    assert not par2.two_streams and rep2["order"].startswith("one stream"), (rank, rep2)
    return par, full, mine
