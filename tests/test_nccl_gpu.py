"""The multi-GPU transport proper: torch.distributed backend "nccl" (= RCCL over xGMI on ROCm), world_size 2, one process
per GPU.  Needs TWO GPUs: on the 1-GPU boxes of this build it SKIPS (the same layout, exchanges and kernels are covered there
by test_sharded_gpu.py with virtual ranks on one GPU and by test_parallel_gloo.py on CPU); on a multi-GPU node it checks
  * every frame-shard exchange primitive on device tensors (halo p2p, fp64 GroupNorm all-reduce, the in-place asynchronous
    all_gather_into_tensor of K|V with its key mask, the compacting gather), and the masked temporal-attention kernel on
    the gathered buffer against attention over the whole clip;
  * the whole pipeline in the 2-rank product layout (one CFG half per GPU) against the single-rank latents.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
T, HW, HEADS = 5, 12, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ref_latents):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from frame_exchange_checks import check_frame_exchanges
        from mofa_video_amd import ops
        dev = torch.device("cuda", rank)
        Cc = HEADS * 64
        par, full, mine = check_frame_exchanges(rank, world, T, HW, 3 * Cc, dev)
        lay = par.lay
        whole = ops.attn_temporal(full[:, :Cc], full[:, Cc:2 * Cc], full[:, 2 * Cc:], 1, T, HW, HEADS)
        buf, own = par.kv_buffer(HW, 2 * Cc, dev)
        ops.copy2d(mine[:, Cc:], own)
        work = par.kv_gather_begin(buf, HW)
        q = mine[:, :Cc].contiguous()                         # (the Q projection runs here in the product)
        work.wait()
        part = ops.attn_temporal(q, buf[:, :Cc], buf[:, Cc:], 1, par.kv_slots, HW, HEADS, Tq=lay.T_loc, key_mask=par.kv_mask)
        err = (part.float() - whole[lay.f0 * HW:lay.f1 * HW].float()).abs().max().item()
        assert err < 2e-3, err

        # the product layout on 2 ranks: one CFG half per GPU, noise predictions swapped per step
        from test_sharded_gpu import H, STEPS, W, build_run
        from mofa_video_amd.parallel import FrameParallel, Layout, TorchComm
        run = build_run(dev)
        par2 = FrameParallel(Layout(world, rank, 4), TorchComm(lambda r: Layout(world, r, 4)))
        lat = run(par2, "latent", 4)
        ref = ref_latents.to(dev)
        e = ((lat.float() - ref.float()).norm() / ref.float().norm()).item()
        assert e < 2e-3, e
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_nccl_world2_frame_exchanges_and_cfg_pair():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL world_size 2); covered by virtual ranks + gloo on 1-GPU boxes")
    import torch.multiprocessing as mp
    from test_sharded_gpu import build_run
    ref = build_run("cuda:0")(None, "latent", 4).cpu()
    mp.spawn(_worker, args=(2, _free_port(), ref), nprocs=2, join=True)
