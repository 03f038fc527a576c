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

    ext = par.halo(mine, HW)                                              # neighbour frames / zeros at the clip ends
    exp_prev = full[(lay.f0 - 1) * HW:lay.f0 * HW] if lay.f0 > 0 else torch.zeros_like(full[:HW])
    exp_next = full[lay.f1 * HW:(lay.f1 + 1) * HW] if lay.f1 < T else torch.zeros_like(full[:HW])
    assert torch.equal(ext[:HW], exp_prev) and torch.equal(ext[(lay.T_loc + 1) * HW:], exp_next)
    assert torch.equal(ext[HW:(lay.T_loc + 1) * HW], mine)

    sums = torch.stack([mine.double().sum(0), (mine.double() ** 2).sum(0)], -1)[:32].reshape(1, -1, 2).clone()
    red = par.reduce_gn(sums)
    parts = [full[a * HW:b * HW] for a, b in lay.bounds]
    fs = sum(torch.stack([p.double().sum(0), (p.double() ** 2).sum(0)], -1)[:32].reshape(1, -1, 2) for p in parts)
    assert torch.allclose(red, fs, rtol=1e-12)

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
    return par, full, mine
