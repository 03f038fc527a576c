"""Does running the MOFA-Adapter's ControlNet trunk and the UNet's encoder half of one denoise step on two HIP streams pay?
The two are independent until the residuals are added: every launch is a grid of persistent
one-per-CU workgroups, so a second stream's kernel can only take the CUs the first one's tail round leaves idle -- the probe
measures whether filling those tails (and the launch gaps) beats the extra cache pressure.  Serial and overlapped steps alternate
in one process; the noise predictions must be bit-identical.

    python tools/two_stream_probe.py [steps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mofa_video_amd import lib, ops  # noqa: E402
from mofa_video_amd.blocks import Ctx  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    lib.load()
    pipe = bench.build_pipeline(dev)
    inp = bench.synthetic_inputs(dev)
    unet, cn = pipe.unet, pipe.controlnet
    T, H, W = bench.T, bench.H, bench.W
    h, w = H // 8, W // 8
    emb = inp["image_embeddings"].to(dev)
    emb = torch.cat([torch.zeros_like(emb), emb], 0)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)
    warped = cn.prepare_condition(inp["cond"][:1].to(dev), inp["flow"][:1].to(dev))
    x = torch.zeros((2 * T * h * w, max(unet.in_ld, cn.in_ld)), dtype=torch.float16, device=dev)
    x[:, :8] = torch.randn(x.shape[0], 8, device=dev).half()
    c_cn, c_un = Ctx(2, T), Ctx(2, T)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def step(overlap):
        cn.make_ctx(1.5, emb, ids, 2, T, base=c_cn)
        unet.make_ctx(1.5, emb, ids, 2, T, base=c_un)
        if not overlap:
            down, mid = cn.forward_tokens(x, c_cn, h, w, warped, 1.0)
            enc = unet.encode_tokens(x, c_un, h, w)
        else:
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                down, mid = cn.forward_tokens(x, c_cn, h, w, warped, 1.0)
            enc = unet.encode_tokens(x, c_un, h, w)
            main_s.wait_stream(side)
            for t in list(down) + [mid]:
                t.record_stream(main_s)
        return unet.decode_tokens(enc, c_un, down, mid)

    rows = T * h * w
    cc = [(Ctx(1, T), Ctx(1, T)) for _ in range(2)]
    st4 = [torch.cuda.Stream() for _ in range(4)]

    def step_halves(four):
        """each CFG half as its own stream of launches (half the rows per launch; the other half's launches fill its tail
        rounds); four: the half's ControlNet trunk and UNet encoder on separate streams as well"""
        outs = [None, None]
        for s_ in st4:
            s_.wait_stream(main_s)
        for hf in range(2):
            c_c, c_u = cc[hf]
            xh = x[hf * rows:(hf + 1) * rows]
            sa, sb = st4[hf], st4[2 + hf] if four else st4[hf]
            with torch.cuda.stream(sa):
                cn.make_ctx(1.5, emb, ids, 1, T, base=c_c, half=hf)
                down, mid = cn.forward_tokens(xh, c_c, h, w, warped, 1.0)
            with torch.cuda.stream(sb):
                unet.make_ctx(1.5, emb, ids, 1, T, base=c_u, half=hf)
                enc = unet.encode_tokens(xh, c_u, h, w)
                sb.wait_stream(sa)
                for t in list(down) + [mid]:
                    t.record_stream(sb)
                outs[hf] = unet.decode_tokens(enc, c_u, down, mid)
                outs[hf].record_stream(main_s)
        for s_ in st4:
            main_s.wait_stream(s_)
        return torch.cat(outs, 0)

    cd = [Ctx(1, T), Ctx(1, T)]

    def step_mixed():
        """ControlNet || UNet encoder on the full CFG batch, then the UNet decoder as two half-batch streams"""
        cn.make_ctx(1.5, emb, ids, 2, T, base=c_cn)
        unet.make_ctx(1.5, emb, ids, 2, T, base=c_un)
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            down, mid = cn.forward_tokens(x, c_cn, h, w, warped, 1.0)
        sample, skips, counts, H2, W2 = unet.encode_tokens(x, c_un, h, w)
        main_s.wait_stream(side)
        for t in list(down) + [mid]:
            t.record_stream(main_s)
        outs = [None, None]
        st = [main_s, side]
        side.wait_stream(main_s)
        for hf in range(2):
            with torch.cuda.stream(st[hf]):
                unet.make_ctx(1.5, emb, ids, 1, T, base=cd[hf], half=hf)
                half_rows = lambda t_: t_[hf * (t_.shape[0] // 2):(hf + 1) * (t_.shape[0] // 2)]
                enc_h = (half_rows(sample), [half_rows(k) for k in skips], counts, H2, W2)
                outs[hf] = unet.decode_tokens(enc_h, cd[hf], [half_rows(r) for r in down], half_rows(mid))
        main_s.wait_stream(side)
        outs[1].record_stream(main_s)
        return torch.cat(outs, 0)

    modes = {"one stream": lambda: step(False), "CN || encoder": lambda: step(True), "CN||enc, dec0||dec1": step_mixed,
             "half || half": lambda: step_halves(False), "4 streams": lambda: step_halves(True)}
    ref = step(False).clone()
    torch.cuda.synchronize()
    for name, fn in modes.items():
        out = fn().clone()
        torch.cuda.synchronize()
        d = (out.float() - ref.float())
        print(f"{name:14s}: equal {bool(torch.equal(out, ref))}  rel-L2 to one stream {float(d.norm() / ref.float().norm()):.2e}")
    res = {k: [] for k in modes}
    for _ in range(steps):
        for name, fn in modes.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1))
    for name, v in res.items():
        print(f"{name:14s} ms per step:", " ".join(f"{t:.1f}" for t in v), " median", f"{sorted(v)[len(v) // 2]:.1f}")


if __name__ == "__main__":
    main()
