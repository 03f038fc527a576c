#!/usr/bin/env python
"""Headline benchmark: denoised frames/sec, 25 f 576x1024 SVD + MOFA-Adapter, 25 denoise steps (+ VAE decode),
on N MI355X (BASELINE.json metric; config[1] "MOFA-Video-Traj, 25-frame 576x1024, 25 steps, single trajectory
hint").

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: spawns the N ranks itself (spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same job under the driver's launcher (WORLD_SIZE must equal N)

One "step" = one whole clip, the reference pipeline call from the conditioning image to the frames: image conditioning
(antialiased resize + CLIP ViT-H/14 image encoder, noise augmentation + VAE encoder; once per clip, < 1 % of the time) +
adapter condition preparation + 25 denoise steps (ControlNet trunk + UNet + CFG/Euler) + chunked temporal-VAE decode, on
synthetic inputs already resident in HBM.  Weights are seeded random in the
reference checkpoint layout (no checkpoints offline), fp16 storage / fp32 accumulate -- the reference's precision.
N > 1 (one process per GPU): default ``--mode shard`` partitions ONE clip over the ranks -- 2-way CFG x N/2 frame
shards with RCCL exchanges only at the temporal ops (mofa_video_amd/parallel.py; strong scaling) -- and
``--mode replicas`` runs one independent clip per rank (weak scaling).  The barrier + max-over-ranks timing contract
is kept in both.  Rank 0 prints ONE JSON line with the extra
``roofline`` (dominant kernel = MFMA implicit-GEMM, timed per launch with HIP events on the launch stream during
the timed region) and ``cpu_baseline`` (the CPU oracle on a bounded sample) objects.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, H, W, STEPS, CHUNK = 25, 576, 1024, 25, 8
FUNCTIONAL_ONLY = False
if os.environ.get("MOFA_BENCH_DENOISE_STEPS"):   # functional checks only -- a line produced with this set is not a result: it
    STEPS = int(os.environ["MOFA_BENCH_DENOISE_STEPS"])   # says "functional_only": true and names the real step count
    FUNCTIONAL_ONLY = STEPS != 25
MFMA_PEAK_TFLOPS = 2500.0    # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


def synthetic_inputs(device, seed=42):
    """one Gaussian-bump trajectory growing linearly to (+64, +32) px."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    h, w = H // 8, W // 8
    lat = torch.randn(1, T, 4, h, w, generator=g)
    il = torch.randn(1, 4, h, w, generator=g) / 0.18215
    emb = torch.randn(1, 1, 1024, generator=g)
    cond = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    image = cond * 0.5 + 0.5                                  # the conditioning image in [0, 1] (= the first frame)
    ys = torch.arange(H, dtype=torch.float32).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32).view(1, W)
    sig = 0.15 * min(H, W)
    bump = torch.exp(-((xs - W / 2) ** 2 + (ys - H / 2) ** 2) / (2 * sig * sig))
    flow = torch.zeros(1, T - 1, 2, H, W)
    for i in range(T - 1):
        f = (i + 1) / (T - 1)
        flow[0, i, 0] = bump * 64.0 * f
        flow[0, i, 1] = bump * 32.0 * f
    return {k: v.to(device) for k, v in dict(latents=lat, image_latents=il, image_embeddings=emb, cond=cond,
                                             flow=flow, image=image).items()}


def build_pipeline(device, seed=0, frontend=False):
    """frontend=True also builds the CLIP image encoder and the VAE encoder half (the once-per-clip conditioning)"""
    from mofa_video_amd import schema
    from mofa_video_amd.adapter import FlowControlNet
    from mofa_video_amd.clip import CLIPVisionModelWithProjection
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    mods = []
    vae_schema = schema.vae_decoder_schema()
    todo = [(UNetSpatioTemporalConditionControlNetModel, schema.unet_schema()), (FlowControlNet, schema.controlnet_schema()),
            (AutoencoderKLTemporalDecoder, vae_schema)]
    if frontend:
        vae_schema.update(schema.vae_encoder_schema())
        todo.append((CLIPVisionModelWithProjection, schema.clip_vision_schema()))
    for i, (cls, sch) in enumerate(todo):
        sd = schema.synthetic_state_dict(sch, seed=seed + i, device=device)
        mods.append(cls(sd, None, device))
        del sd
        torch.cuda.empty_cache()
    unet, cn, vae = mods[:3]
    return FlowControlNetPipeline(vae=vae, image_encoder=mods[3] if frontend else None, unet=unet, controlnet=cn,
                                  scheduler=EulerDiscreteScheduler())


def synthetic_landmark_inputs(device, n_frames=T, seed=43):
    """68 landmarks on a face-sized ellipse, every point drifting <= 20 px over the clip; the
    dense flow is 68 Gaussian bumps (sigma 12 px) carrying those displacements (what CMP produces from the sparse landmark
    flow), the pose images are the reference's polyline drawing of the landmarks (mofa_video_amd/landmarks.py)."""
    import numpy as np
    from mofa_video_amd.landmarks import pose_images
    g = torch.Generator(device="cpu").manual_seed(seed)
    th = torch.linspace(0, 2 * 3.14159265, 69)[:68]
    base = torch.stack([W / 2 + 0.16 * W * torch.cos(th), H / 2 + 0.30 * H * torch.sin(th)], 1)      # [68, 2] (x, y)
    drift = (torch.rand(68, 2, generator=g) * 2 - 1) * 20.0
    lm = torch.stack([base + drift * (f / max(n_frames - 1, 1)) for f in range(n_frames)])           # [N, 68, 2]
    ys = torch.arange(H, dtype=torch.float32, device=device).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32, device=device).view(1, W)
    flow = torch.zeros(1, n_frames - 1, 2, H, W, device=device)
    wsum = torch.zeros(H, W, device=device)
    bumps = []
    for k in range(68):
        b = torch.exp(-((xs - float(base[k, 0])) ** 2 + (ys - float(base[k, 1])) ** 2) / (2 * 12.0 ** 2))
        bumps.append(b)
        wsum += b
    for i in range(n_frames - 1):
        fx = torch.zeros(H, W, device=device)
        fy = torch.zeros(H, W, device=device)
        for k in range(68):
            d = lm[i + 1, k] - lm[0, k]
            fx += bumps[k] * float(d[0])
            fy += bumps[k] * float(d[1])
        flow[0, i, 0], flow[0, i, 1] = fx / wsum.clamp_min(1.0), fy / wsum.clamp_min(1.0)
    return dict(flow=flow, landmarks=pose_images(np.asarray(lm), H, W).to(device))


def synthetic_mask(device):
    """config 4: binary mask [1,1,H,W], a centred rectangle covering 25 % of the area (1 = landmark adapter)"""
    m = torch.zeros(1, 1, H, W, device=device)
    m[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
    return m


def build_config_pipeline(device, config, seed=0):
    """configs 3 / 4 / 5 reuse config 2's UNet / VAE / front end and add the landmark adapter (+ keep the trajectory one)"""
    from mofa_video_amd import schema
    from mofa_video_amd.adapter import LandmarkFlowControlNet
    from mofa_video_amd.pipeline import HybridFlowControlNetPipeline, KeypointFlowControlNetPipeline
    base = build_pipeline(device, seed=seed, frontend=True)
    if config == 2:
        return base
    sd = schema.synthetic_state_dict(schema.ldmk_controlnet_schema(), seed=seed + 7, device=device)
    face = LandmarkFlowControlNet(sd, None, device)
    del sd
    torch.cuda.empty_cache()
    common = dict(vae=base.vae, image_encoder=base.image_encoder, unet=base.unet, scheduler=base.scheduler)
    if config == 3:
        return KeypointFlowControlNetPipeline(controlnet=face, **common)
    if config == 4:
        return HybridFlowControlNetPipeline(face_controlnet=face, drag_controlnet=base.controlnet, **common)
    return KeypointFlowControlNetPipeline(controlnet=face, drag_controlnet=base.controlnet, **common)


LONG_FRAMES = 97      # config 5: "4 x 25-frame chunks": window 25, stride 12 -> 7 full windows cover 97 frames


def config_inputs(device, config, seed=42):
    inp = synthetic_inputs(device, seed)
    if config == 2:
        return inp
    n = LONG_FRAMES if config == 5 else T
    lmk = synthetic_landmark_inputs(device, n)
    if config == 5:                                           # the trajectory hint of config 2 stretched over the long clip
        g = torch.Generator(device="cpu").manual_seed(seed)
        inp["latents"] = torch.randn(1, n, 4, H // 8, W // 8, generator=g).to(device)
        f = torch.arange(1, n, device=device, dtype=torch.float32).view(1, n - 1, 1, 1, 1) / (n - 1)
        inp["drag_flow"] = inp["flow"][:, -1:].repeat(1, n - 1, 1, 1, 1) * f
    else:
        inp["drag_flow"] = inp["flow"]
    inp["flow"], inp["landmarks"] = lmk["flow"], lmk["landmarks"]
    inp["mask"] = synthetic_mask(device)
    return inp


def run_config(pipe, inp, config):
    """one clip of BASELINE.json configs[config - 1] through the reference call of that configuration"""
    gen = torch.Generator().manual_seed(1234)
    common = dict(controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W,
                  num_inference_steps=STEPS, decode_chunk_size=CHUNK, latents=inp["latents"], output_type="pt", generator=gen)
    if config == 2:
        return run_clip(pipe, inp)
    if config == 3:      # Keypoint, 25 frames = one window (MOFA-Video-Keypoint/mofa_keypoint.py:346-360)
        return pipe(inp["image"], landmarks=inp["landmarks"], window_size=T, stride=T // 2, num_frames=T, **common).frames
    if config == 4:      # Hybrid (MOFA-Video-Hybrid/run_gradio_audio_driven.py:452-471)
        return pipe(inp["image"], landmarks=inp["landmarks"], drag_flow=inp["drag_flow"], mask=inp["mask"], num_frames=T,
                    **common).frames
    return pipe(inp["image"], landmarks=inp["landmarks"], window_size=T, stride=T // 2, num_frames=LONG_FRAMES,
                drag_flow=inp["drag_flow"], mask=inp["mask"], **common).frames


WORKLOADS = {
    2: "MOFA-Video-Traj, 25-frame 576x1024, 25 denoise steps + temporal VAE decode (chunk 8), single trajectory hint, "
       "SVD-XT UNet + MOFA-Adapter, CFG 1->3, seeded random weights in the reference checkpoint layout",
    3: "MOFA-Video-Keypoint, 25-frame 576x1024, 25 denoise steps + temporal VAE decode (chunk 8), landmark-driven dense flow "
       "(68 landmarks, polyline pose images), SVD-XT UNet + landmark MOFA-Adapter (occlusion matting), seeded random weights",
    4: "MOFA-Video-Hybrid, 25-frame 576x1024, 25 denoise steps + temporal VAE decode (chunk 8), trajectory + landmark "
       "dual adapter blended by a 25 % mask, SVD-XT UNet, seeded random weights",
    5: "Long video: 97 frames (4 x 25-frame chunks: 7 windows of 25, stride 12) 576x1024, 25 denoise steps, hybrid control "
       "in every window, overlap-averaged latents, VAE decode (chunk 8) overlapped with the last step, seeded random weights",
}
# single adapter 218.58 TFLOP per step + 173.57 decode; dual adapter 277.27 per step
CLIP_TFLOPS = {2: 5638.0, 3: 5638.0, 4: 7105.0, 5: 7 * 25 * 277.27 + 173.57 * LONG_FRAMES / 25.0}


def run_clip(pipe, inp):
    """the reference call (MOFA-Video-Traj/run_gradio.py:330-352 -> pipeline.py:293): image in, frames out"""
    out = pipe(inp["image"], controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W, num_frames=T,
               num_inference_steps=STEPS, decode_chunk_size=CHUNK, latents=inp["latents"], output_type="pt",
               generator=torch.Generator().manual_seed(1234))
    return out.frames


def _cpu_baseline_worker():
    """Runs in a child process (bounded by a timeout in cpu_baseline()).  Prints one JSON line."""
    from torch.utils.flop_counter import FlopCounterMode
    from oracle.controlnet import FlowControlNet
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    t0 = time.time()
    with torch.device("meta"):
        u, c = UNetSpatioTemporalConditionControlNetModel(), FlowControlNet()
    u, c = u.to_empty(device="cpu"), c.to_empty(device="cpu")
    with torch.no_grad():
        for m in (u, c):
            for name, p in m.named_parameters():
                if p.dim() > 1:            # cheap deterministic non-trivial fill (timing is data independent)
                    n = p.numel()
                    fan = p[0].numel()
                    p.view(-1).copy_((torch.arange(n, dtype=torch.float32) % 251 - 125.0) * (fan ** -0.5 / 125.0))
                elif name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
    setup = time.time() - t0
    g = torch.Generator().manual_seed(0)
    Tc, Hc, Wc = 8, 256, 256
    x = torch.randn(2, Tc, 8, Hc // 8, Wc // 8, generator=g)
    emb = torch.randn(2, 1, 1024, generator=g)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    cond = torch.rand(2, 3, Hc, Wc, generator=g)
    flow = torch.randn(2, Tc - 1, 2, Hc, Wc, generator=g)
    fc = FlopCounterMode(display=False)
    t0 = time.time()
    with torch.no_grad(), fc:
        dr, mr, _, _ = c(x, torch.tensor(1.0), emb, ids, controlnet_cond=cond, controlnet_flow=flow, return_dict=False)
        u(x, torch.tensor(1.0), emb, down_block_additional_residuals=dr, mid_block_additional_residual=mr,
          return_dict=False, added_time_ids=ids)
    dt = time.time() - t0
    tflop = fc.get_total_flops() / 1e12
    del u, c, dr, mr
    # second part of the sample: the temporal VAE decoder at the FULL 576x1024 resolution on a 2-frame chunk (its
    # work is linear in the frames of a chunk; a whole 8-frame chunk is 54 TFLOP, minutes of host time)
    from oracle.vae import AutoencoderKLTemporalDecoder
    t0 = time.time()
    with torch.device("meta"):
        v = AutoencoderKLTemporalDecoder()
    v = v.to_empty(device="cpu")
    with torch.no_grad():
        for name, p in v.named_parameters():
            if p.dim() > 1:
                n = p.numel()
                p.view(-1).copy_((torch.arange(n, dtype=torch.float32) % 251 - 125.0) * (p[0].numel() ** -0.5 / 125.0))
            elif name.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    setup += time.time() - t0
    z = torch.randn(2, 4, H // 8, W // 8, generator=g)
    fcv = FlopCounterMode(display=False)
    t0 = time.time()
    with torch.no_grad(), fcv:
        v.decode(z, num_frames=2)
    dtv = time.time() - t0
    print(json.dumps(dict(cores=cores, avail=avail, setup=setup, dt=dt, tflop=tflop, dt_vae=dtv,
                          tflop_vae=fcv.get_total_flops() / 1e12)))


def _cpu_baseline_full_worker():
    """--cpu-baseline-full: ONE denoise step of the oracle at the FULL configuration (25 f 576x1024, CFG batch 2: adapter +
    ControlNet trunk + UNet, 218.58 TFLOP by the work model) on the host cores, timed once (SURVEY 8d's own prescription; the
    default bench uses the bounded 8 f x 256x256 sample instead because this takes minutes).  Prints one JSON line."""
    from oracle.controlnet import FlowControlNet
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("MOFA_CPU_BASELINE_THREADS", "64"))))
    torch.set_num_threads(cores)
    with torch.device("meta"):
        u, c = UNetSpatioTemporalConditionControlNetModel(), FlowControlNet()
    u, c = u.to_empty(device="cpu"), c.to_empty(device="cpu")
    with torch.no_grad():
        for m in (u, c):
            for name, p in m.named_parameters():
                if p.dim() > 1:
                    n = p.numel()
                    p.view(-1).copy_((torch.arange(n, dtype=torch.float32) % 251 - 125.0) * (p[0].numel() ** -0.5 / 125.0))
                elif name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, T, 8, H // 8, W // 8, generator=g)
    emb = torch.randn(2, 1, 1024, generator=g)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    cond = torch.rand(2, 3, H, W, generator=g)
    flow = torch.randn(2, T - 1, 2, H, W, generator=g)
    t0 = time.time()
    with torch.no_grad():
        dr, mr, _, _ = c(x, torch.tensor(1.0), emb, ids, controlnet_cond=cond, controlnet_flow=flow, return_dict=False)
        t1 = time.time()
        u(x, torch.tensor(1.0), emb, down_block_additional_residuals=dr, mid_block_additional_residual=mr, return_dict=False,
          added_time_ids=ids)
    t2 = time.time()
    print(json.dumps(dict(cores=cores, avail=avail, adapter_s=t1 - t0, unet_s=t2 - t1, step_s=t2 - t0, step_tflop=218.58)))


def cpu_baseline_full(timeout=3000):
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import bench; bench._cpu_baseline_full_worker()"], cwd=ROOT, capture_output=True,
                       text=True, timeout=timeout)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rate = d["step_tflop"] / d["step_s"]
    return dict(kind="port", cores=d["cores"], unit="denoised frames/sec",
                value=25.0 / (25 * d["step_s"] + 173.57 / 1.06),
                sample=(f"oracle (fp32 torch CPU, {d['cores']} threads of {d['avail']}): ONE full-size denoise step, 25 f 576x1024, CFG 2 "
                        f"(adapter + ControlNet trunk {d['adapter_s']:.0f} s, UNet {d['unet_s']:.0f} s = {d['step_s']:.0f} s for 218.58 TFLOP "
                        f"= {rate:.3f} TFLOP/s); value = 25 frames / (25 x that step + 173.57 TFLOP of decode at the 1.06 TFLOP/s of "
                        "the default sample's VAE part)"))


def cpu_baseline(timeout=420):
    """The CPU oracle (fp32 PyTorch restatement of the reference pipeline, oracle/) on a BOUNDED sample of the same
    workload, two parts: ONE denoise step (MOFA-Adapter/ControlNet + UNet, CFG batch 2) of the full-size SVD-XT architecture
    at 8 frames x 256x256 (BASELINE config[0] geometry), and the temporal VAE decoder on a 2-frame chunk at the full 576x1024;
    FLOPs counted by torch's FlopCounterMode, each part converted to seconds per clip through the analytic work model
    (218.58 TFLOP per denoise step, 173.57 TFLOP of decode at 25 f 576x1024) with ITS OWN measured rate.  Runs in
    a child process with a hard timeout so the default bench stays within minutes on any host."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import bench; bench._cpu_baseline_worker()"], cwd=ROOT,
                           capture_output=True, text=True, timeout=timeout)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        return dict(value=None, unit="denoised frames/sec", cores=0, kind="port",
                    sample=f"CPU oracle sample did not finish within {timeout} s ({type(e).__name__})")
    # whole-clip time on the host = 25 steps x (218.58 TFLOP per step at the UNet sample's rate) + 173.57 TFLOP of decode at the
    # VAE sample's rate (SURVEY 8d work model for 25 f 576x1024, single adapter)
    r_unet, r_vae = d["tflop"] / d["dt"], d["tflop_vae"] / d["dt_vae"]
    clip_s = 25 * 218.58 / r_unet + 173.57 / r_vae
    return dict(value=25.0 / clip_s, unit="denoised frames/sec", cores=d["cores"], kind="port",
                sample=(f"oracle (fp32 torch CPU, {d['cores']} threads of {d['avail']} available), two parts: (1) one denoise step of "
                        f"the full-size SVD-XT UNet + MOFA ControlNet at 8 f x 256x256, CFG batch 2 = {d['tflop']:.3f} TFLOP "
                        f"(FlopCounterMode) in {d['dt']:.1f} s = {r_unet:.4f} TFLOP/s; (2) the temporal VAE decoder on a 2-frame chunk "
                        f"at the full 576x1024 = {d['tflop_vae']:.3f} TFLOP in {d['dt_vae']:.1f} s = {r_vae:.4f} TFLOP/s (model set-up "
                        f"{d['setup']:.0f} s untimed); value = 25 frames / (25 x 218.58 TFLOP / rate 1 + 173.57 TFLOP / rate 2) = "
                        f"25 / {clip_s:.0f} s.  The full-size figure exists too: ONE denoise step of this oracle at 25 f 576x1024 = 294 s "
                        f"on 64 threads (0.744 TFLOP/s) => 0.0033 frames/s, profiles/r04b_cpu_baseline_full.json (python bench.py "
                        f"--cpu-baseline-full, several minutes)"))


def spawn_command(n, argv, port):
    """The command line `python bench.py --gpus N ...` re-executes itself under: one process per GPU, torch.distributed.run on
    the loopback address (the driver's own launch line for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py"), *argv]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks here and pass rank 0's single JSON
    line through.  Fewer than N visible GPUs is an error -- never a fall-back to one GPU -- unless MOFA_BENCH_ONE_GPU=1
    (functional check of the multi-process path: all ranks on GPU 0, use --backend gloo)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("MOFA_BENCH_ONE_GPU") != "1":
        raise SystemExit(f"bench.py: --gpus {n} requested but {have} GPU(s) visible; refusing to run a smaller job under an "
                         f"{n}-GPU command (MOFA_BENCH_ONE_GPU=1 --backend gloo runs the {n}-process path on one GPU as a functional check)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(spawn_command(n, argv, port), env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="no GPU work: time ONE full-size denoise step of the CPU oracle (25 f 576x1024; several minutes) and print "
                         "the cpu_baseline object it yields (kept under profiles/)")
    ap.add_argument("--no-launch-timer", action="store_true",
                    help="A/B switch: do not bracket the launches with HIP events (the line then carries no live roofline leg); "
                         "measures what the event records cost inside the timed region")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[config - 1]: 2 = Traj (the headline metric's configuration, default), "
                         "3 = Keypoint, 4 = Hybrid, 5 = 97-frame long video with hybrid control (about a minute per clip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                    "single-GPU functional check of the multi-process path, see tests/README in DESIGN.md)")
    ap.add_argument("--mode", choices=["shard", "replicas"], default="shard",
                    help="N > 1: 'shard' = ONE clip per step partitioned over the ranks (2-way CFG x N/2 frame shards, "
                         "RCCL exchanges at the temporal ops; strong scaling) or 'replicas' = one independent clip per "
                         "rank (no data-path collective; weak scaling)")
    ap.add_argument("--single-stream", action="store_true",
                    help="every clip in single-stream order (pipeline.overlap_adapter = False): the mode the roofline leg and the "
                         "rocprofv3 kernel statistics are taken in -- kernel durations are exclusive only when nothing runs beside")
    ap.add_argument("--graph-steps", type=int, default=-1, help="A/B switch: pipeline.graph_steps = 0 | 1 (one captured hipGraph per "
                    "clip replayed for steps 1 .. 24; default: the pipeline's)")
    ap.add_argument("--split-decoder", type=int, default=-1, help="A/B switch: pipeline.split_decoder = 0 | 1 (default: the pipeline's)")
    ap.add_argument("--gn-stats", type=int, default=-1, help="A/B switch: ops.GN_STATS = 0 | 1 (GroupNorm partial sums from the "
                    "producing implicit-GEMM epilogue; default: on)")
    ap.add_argument("--ff-fused", type=int, default=-1, help="A/B switch: ops.FF_FUSED = 0 | 1 (the level-0 feed-forwards as one fused launch "
                    "each, or LayerNorm + two implicit GEMMs; default: on)")
    ap.add_argument("--lib", default="", help="A/B switch: load this build of libmofa_hip.so instead of the in-tree one "
                    "(same-box comparison of two kernel builds; the path is echoed in config.library)")
    args = ap.parse_args()
    if args.cpu_baseline_full:
        print(json.dumps(cpu_baseline_full()))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))    # plain `python bench.py --gpus N`: N ranks, never a silent 1-GPU run
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or plain `python bench.py --gpus N`, which spawns the ranks itself)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MOFA_BENCH_ONE_GPU") == "1":      # functional check: all ranks share GPU 0 (gloo transport)
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from mofa_video_amd import lib, ops
    if args.lib:
        lib.LIB_PATH = os.path.abspath(args.lib)
    lib.load()                                              # fails loudly without the HIP library
    cfg = args.config
    pipe = build_config_pipeline(dev, cfg, seed=0)
    mode = args.mode if world > 1 else "single"
    comm_paths = window_plan = None
    if mode == "shard":
        # every rank must see the same clip; partition = mofa_video_amd/parallel.py.  A failure here is an error: a
        # strong-scaling request must never silently turn into independent replicas
        from mofa_video_amd.parallel import (FrameParallel, GroupedWindowParallel, Layout, TorchComm, WindowParallel, plan_windows,
                                             window_layout_costs)
        if cfg == 5:                                        # long video: the layout the cost table picks for (windows, ranks)
            from mofa_video_amd.pipeline import window_views
            nwin = len(set(window_views(LONG_FRAMES, T, T // 2)))
            plan = plan_windows(nwin, world)
            window_plan = dict(windows=nwin, chosen=plan, costs=[(round(c, 3), n, G, g) for c, n, G, g in window_layout_costs(nwin, world)])
            if plan[0] == "window":
                comm = TorchComm(lambda r: Layout(1, 0, T))
                pipe.parallel = WindowParallel(comm, rank, world)
            elif plan[0] == "frame":
                comm = TorchComm(lambda r: Layout(world, r, T))
                pipe.parallel = FrameParallel(Layout(world, rank, T), comm)
            else:
                comm = TorchComm(GroupedWindowParallel.layout_of_rank(world, plan[2], T))
                pipe.parallel = GroupedWindowParallel(comm, rank, world, plan[2], T)
            comm.all_gather_world(torch.ones(4, device=dev))
        else:
            if world % 2 != 0:
                raise ValueError("shard mode needs an even number of ranks (2-way CFG split)")
            comm = TorchComm(lambda r: Layout(world, r, T))
            pipe.parallel = FrameParallel(Layout(world, rank, T), comm)
            probe = torch.ones(4, device=dev, dtype=torch.float64)          # exercise the collectives once
            comm.all_reduce_sum(probe, pipe.parallel.lay.frame_group)
            comm.all_gather(probe, pipe.parallel.lay.pair_group)
            comm_paths = pipe.parallel.self_check(dev)                      # fast paths verified on THIS transport, or switched off
        torch.cuda.synchronize()
    inp = config_inputs(dev, cfg, seed=42 + (rank if mode == "replicas" else 0))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pipe.overlap_adapter = not args.single_stream
    if args.split_decoder >= 0:
        pipe.split_decoder = bool(args.split_decoder)
    if args.gn_stats >= 0:
        from mofa_video_amd import ops as _ops
        _ops.GN_STATS = bool(args.gn_stats)
    if args.graph_steps >= 0:
        pipe.graph_steps = bool(args.graph_steps)
    if args.ff_fused >= 0:
        from mofa_video_amd import ops as _ops
        _ops.FF_FUSED = bool(args.ff_fused)
    for _ in range(args.warmup):
        run_config(pipe, inp, cfg)
    # Roofline leg: HIP events around every implicit-GEMM / attention / softsplat launch of the LAST clip of the timed region
    # (on the launch stream), and that clip runs in SINGLE-STREAM order.  Two reasons, both measured:
    #  * bracketing all K clips costs 2.3 % of the clip time (two event records per launch x 19 000 timed launches per clip:
    #    6 674 against 6 525 ms, profiles/archive/r03c_bench_{timer,notimer}.log), which `value` would carry;
    #  * the pipeline enqueues the adapter's ControlNet trunk and the UNet's encoder half on two HIP streams (pipeline.py,
    #    _denoise_forward: 3.0-4.4 % of a step).  A launch's event (or rocprofv3) duration then includes the time it waits for
    #    the CUs the other stream's kernel holds, so it no longer measures the kernel: the per-launch durations are taken with
    #    nothing running beside (the same mode as profiles/archive/r03*_kernel_stats_bench.md, `--single-stream`).
    # `value` is still all K clips / the whole timed region, the instrumented one included; config.clip_ms has both kinds.
    timer = ops.LaunchTimer()
    pipe.overlap_adapter = not args.single_stream
    clip_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    clip_ev[0].record()
    for i in range(args.steps):
        if i == args.steps - 1 and not args.no_launch_timer:
            ops.TIMER, pipe.overlap_adapter = timer, False
        frames = run_config(pipe, inp, cfg)
        clip_ev[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    pipe.overlap_adapter = not args.single_stream
    clip_ms = [round(clip_ev[i].elapsed_time(clip_ev[i + 1]), 1) for i in range(args.steps)]
    # "pt" output = the reference's tensor2vid result: a list with one [T,3,H,W] tensor in [0,1] per batch element
    # (single rank), or the list of (first_frame, [n,3,H,W]) VAE chunks this rank decoded (shard mode)
    finite = all(bool(torch.isfinite(f[1] if isinstance(f, tuple) else f).all().item()) for f in frames)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        summ = timer.summary()
        ig = summ.get("igemm_f16_kernel", dict(launches=1, seconds=1.0, flops=0.0))
        avg_s = ig["seconds"] / max(ig["launches"], 1)
        ach = ig["flops"] / ig["seconds"] / 1e12
        at = summ.get("attn_spatial_kernel")
        traffic = None       # HBM bytes per launch from the committed PMC passes (not collectable live)
        import glob
        tags = sorted((os.path.basename(f).split("_")[0] for f in glob.glob(os.path.join(ROOT, "profiles", "r*_igemm_traffic.json"))),
                      reverse=True)
        for tag in tags:                                              # newest committed PMC summary (tools/profile_round.sh)
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_igemm_traffic.json")))
                traffic = dict(hbm_bytes_per_launch=round(tj["hbm_bytes_per_launch"]),
                               source=f"profiles/{tag}_igemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                      "passes, gfx950 x2 FETCH correction)")
                break
            except Exception:  # noqa: BLE001
                pass
        roofline = dict(kernel="igemm_f16_kernel (+ ff320_kernel: the level-0 feed-forward GEMM pairs, fused with their LayerNorm / GELU / residuals)", bound="mfma", achieved=round(ach, 1), peak=MFMA_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=round(ach / MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                        launches_per_clip=ig["launches"],
                        timed_with_events="the last of the K timed clips, run in single-stream order (exclusive kernel durations)",
                        avg_launch_us=round(avg_s * 1e6, 1),
                        algorithmic_tflop_per_launch=round(ig["flops"] / max(ig["launches"], 1) / 1e12, 5),
                        share_of_clip_time=round(ig["seconds"] / (clip_ms[-1] * 1e-3), 3))
        if at:
            roofline["attn_spatial_kernel"] = dict(achieved=round(at["flops"] / at["seconds"] / 1e12, 1), unit="TFLOP/s",
                                                   frac=round(at["flops"] / at["seconds"] / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                                   share_of_clip_time=round(at["seconds"] / (clip_ms[-1] * 1e-3), 3))
        # HBM-bound kernels of the adapter on REAL bytes (8 TB/s peak; MI355X_MICROARCH.md): the forward-splat warp
        # (count / scan / fill / sort / gather, timed as one op; bytes = per flow frame and target pixel: C fp16 read + C fp16
        # written + the flow + its CSR entries) and the full-resolution 16 / 32-channel condition-embedding convolutions (rows x
        # (64 padded input channels + N output channels) x 2 B)
        hbm = {}
        ss = summ.get("softsplat_avg")
        if ss and ss["seconds"] > 0:
            hbm["softsplat_avg (ss_count/scan/fill/sort/gather)"] = dict(
                achieved=round(ss["bytes"] / ss["seconds"] / 1e9, 1), peak=8000.0, unit="GB/s",
                frac=round(ss["bytes"] / ss["seconds"] / 8e12, 4), launches=ss["launches"])
        small = [(tag, v) for tag, v in timer.by_tag().items() if tag[4] <= 32 and tag[3] >= 100000]
        if small:
            by = sum(v[0] * tag[3] * (64 + tag[4]) * 2.0 for tag, v in small)
            sec = sum(v[1] for tag, v in small)
            hbm["igemm_f16_kernel, full-resolution 16/32-channel adapter convolutions"] = dict(
                achieved=round(by / sec / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(by / sec / 8e12, 4),
                launches=sum(v[0] for _, v in small))
        if hbm:
            roofline["hbm_kernels"] = hbm
        clips = args.steps * (world if mode == "replicas" else 1)
        nfr = LONG_FRAMES if cfg == 5 else T
        value = nfr * clips / dt
        par_desc = {"single": "1 GPU", "replicas": f"{world} independent clips, one per GPU, no data-path collective",
                    "shard": (f"one clip over {world} GPUs, layout from parallel.window_layout_costs: {window_plan}; one all-gather of "
                              "the stepped window latents per round; VAE chunks dealt over all ranks" if cfg == 5 else
                              f"one clip over {world} GPUs: 2-way CFG x {max(world // 2, 1)}-way frame shards; RCCL: per temporal "
                              "norm + conv one exchange group (raw halo frames p2p + all-gather of the GroupNorm partials), one "
                              "hidden-token all-gather per temporal attention, CFG pair and final latents all-gathers; " +
                              ("trunk || encoder enqueued in lockstep on two streams" if getattr(pipe.parallel, "two_streams", False)
                               else "trunk then encoder on one stream (MOFA_SHARD_TWO_STREAMS=1 opts into the lockstep two-stream order)") +
                              "; VAE chunks round-robin")}[mode]
        line = {
            "metric": (f"denoised frames/sec, 25f 576x1024 SVD+MOFA, {STEPS} steps" if cfg != 5 else
                       f"denoised frames/sec, 97f (4 x 25f windows) 576x1024 SVD+MOFA hybrid, {STEPS} steps") +
                      (" -- FUNCTIONAL CHECK ONLY, not the headline metric (MOFA_BENCH_DENOISE_STEPS)" if FUNCTIONAL_ONLY else ""),
            **({"functional_only": True} if FUNCTIONAL_ONLY else {}),
            "value": round(value, 4),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong" if mode == "shard" else "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": WORKLOADS[cfg], "baseline_config_index": cfg - 1,
                       "num_frames": nfr, "height": H, "width": W, "num_inference_steps": STEPS, "backend": args.backend if world > 1 else None,
                       "decode_chunk_size": CHUNK, "step_definition": "one whole clip from the conditioning image (CLIP + VAE "
                       "encode, adapter prep, 25 denoise steps, VAE decode)", "parallelism": par_desc,
                       "streams": ("single stream" if args.single_stream else
                                   "adapter trunk || UNet encoder, then the decoder's two CFG halves, on two HIP streams; the last timed clip (HIP "
                                   "events) single-stream"),
                       "graph_steps": bool(pipe.graph_steps), **({"gn_stats": bool(args.gn_stats)} if args.gn_stats >= 0 else {}),
                       **({"ff_fused": bool(args.ff_fused)} if args.ff_fused >= 0 else {}),
                       "clip_ms": clip_ms,
                       "output_finite": finite, "comm_paths": comm_paths, **({"library": args.lib} if args.lib else {}),
                       "effective_tflops_per_gpu_reference_work_model": round(CLIP_TFLOPS[cfg] * clips / dt / world, 1)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and cfg == 2:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
