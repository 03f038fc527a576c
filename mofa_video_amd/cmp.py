"""CMP sparse-to-dense motion encoder on the HIP library: the step right before the hot path, once per clip.

Mirrors the reference surface: ``CMP_demo.run(image, sparse, mask)``
(Traj/models/svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine.py:25-62) and ``get_cmp_flow`` / ``get_flow``
(Traj/run_gradio.py:236-277); weights are the state_dict of the reference's ``CMP`` module (modules/cmp.py:6-25, keys in
schema.cmp_schema()).  Every convolution is an implicit-GEMM launch with BatchNorm (eval) folded into weight and bias
and ReLU / residual in the epilogue; pooling, align_corners bilinear resize and the 99-bin softmax expectation are
the kernels of csrc/cmp_ops.hip.  Activations are token-major fp16 [n*H*W, C] with channel counts padded to 64."""
import torch

from . import lib as L
from . import ops
from . import weights as Wt

BN_EPS = 1e-5
NBINS, FMAX = 99, 50


def _r64(c):
    return (c + 63) // 64 * 64


class _Conv:
    """conv (+ folded BatchNorm) (+ ReLU): weight fp16 [Npad4][k*k*Cinpad64], bias fp32."""

    def __init__(self, sd, conv, bn, dev, ksize, stride=1, dil=1, relu=True):
        w = Wt.f32(sd[conv + ".weight"])
        b = Wt.f32(sd[conv + ".bias"]) if conv + ".bias" in sd else torch.zeros(w.shape[0])
        if bn is not None:
            scale = Wt.f32(sd[bn + ".weight"]) / torch.sqrt(Wt.f32(sd[bn + ".running_var"]) + BN_EPS)
            w = w * scale.view(-1, 1, 1, 1)
            b = Wt.f32(sd[bn + ".bias"]) + (b - Wt.f32(sd[bn + ".running_mean"])) * scale
        self.N = w.shape[0]
        self.plain = ksize == 1 and stride == 1
        wp = Wt.pack_linear(w) if self.plain else Wt.pack_conv3x3(w)
        self.w = Wt.pad_rows(wp).to(dev)
        self.b = Wt.pad_rows(b.view(-1, 1)).view(-1).to(dev)
        self.ksize, self.stride, self.dil, self.act = ksize, stride, dil, (L.ACT_RELU if relu else L.ACT_NONE)

    def __call__(self, x, n, H, W, r1=None, out=None):
        """x [n*H*W, ld]; returns (y [n*Ho*Wo, ld_out], Ho, Wo).  Pad columns of y are zero (next conv's K padding)."""
        if self.plain:
            geom, Ho, Wo = ops.PLAIN, H, W
        else:
            geom = ops.conv3x3_geom(H, W, stride=self.stride, ksize=self.ksize, dil=self.dil)
            Ho, Wo = geom.Hout, geom.Wout
        Np = self.w.shape[0]
        if out is None:
            ld = _r64(Np)
            out = (torch.zeros if ld != Np else torch.empty)((n * Ho * Wo, ld), dtype=torch.float16, device=x.device)
        y = ops.igemm(x, self.w, bias=self.b, geom=geom, r1=r1, act=self.act, out=out[:, :Np] if out.shape[1] != Np else out)
        del y
        return out, Ho, Wo


class _Bottleneck:
    def __init__(self, sd, p, dev, stride, dil, has_down):
        # resnet.py:49-86; layer3/4: conv2 dilated + de-strided, downsample de-strided (:118-129)
        self.c1 = _Conv(sd, p + ".conv1", p + ".bn1", dev, 1)
        self.c2 = _Conv(sd, p + ".conv2", p + ".bn2", dev, 3, stride=stride, dil=dil)
        self.c3 = _Conv(sd, p + ".conv3", p + ".bn3", dev, 1, relu=True)      # ReLU after the residual add (epilogue order)
        self.down = _Conv(sd, p + ".downsample.0", p + ".downsample.1", dev, 1, stride=stride, relu=False) if has_down else None

    def __call__(self, x, n, H, W):
        y, _, _ = self.c1(x, n, H, W)
        y, Ho, Wo = self.c2(y, n, H, W)
        res = x if self.down is None else self.down(x, n, H, W)[0]
        y, _, _ = self.c3(y, n, Ho, Wo, r1=res)
        return y, Ho, Wo


class CMP:
    """the reference's ``CMP`` module at the inference configuration (schema.cmp_schema())."""

    def __init__(self, state_dict, device="cuda"):
        sd, dev = state_dict, device
        e = "image_encoder"
        self.conv1 = _Conv(sd, e + ".conv1", e + ".bn1", dev, 7, stride=2)
        self.layers = []
        for li, (blocks, stride, dil) in enumerate(((3, 1, 1), (4, 2, 1), (6, 1, 2), (3, 1, 4)), start=1):
            self.layers.append([_Bottleneck(sd, f"{e}.layer{li}.{b}", dev, stride if b == 0 else 1, dil, b == 0)
                                for b in range(blocks)])
        self.conv5 = _Conv(sd, e + ".conv5", None, dev, 1, relu=False)
        f = "flow_encoder.features"
        self.s0 = _Conv(sd, f + ".0", f + ".1", dev, 5, stride=2)
        self.s4 = _Conv(sd, f + ".4", f + ".5", dev, 3)
        g = "flow_decoder"
        self.dec = {name: [_Conv(sd, f"{g}.{name}.{first + 3 * j}", f"{g}.{name}.{first + 3 * j + 1}", dev, 3) for j in range(3)]
                    for name, first in (("decoder1", 0), ("decoder2", 1), ("decoder4", 1), ("decoder8", 1))}
        self.fusion8 = _Conv(sd, g + ".fusion8.0", g + ".fusion8.1", dev, 3)
        self.skipconv4 = _Conv(sd, g + ".skipconv4.0", g + ".skipconv4.1", dev, 3)
        self.fusion4 = _Conv(sd, g + ".fusion4.0", g + ".fusion4.1", dev, 3)
        self.skipconv2 = _Conv(sd, g + ".skipconv2.0", g + ".skipconv2.1", dev, 3)
        self.fusion2 = _Conv(sd, g + ".fusion2.0", g + ".fusion2.1", dev, 3)
        self.head = _Conv(sd, g + ".head", None, dev, 1, relu=False)
        self.device = dev

    def forward(self, image, sparse):
        """image fp32 [n,3,H,W] (already *2-1), sparse fp32 [n,4,H,W] -> logits fp16 token-major [n*(H/2)*(W/2), 256]
        (198 valid columns), and (H/2, W/2).  modules/cmp.py:27-37."""
        n, _, H, W = image.shape
        assert H % 8 == 0 and W % 8 == 0
        # --- image encoder (resnet.py:152-166)
        x = ops.nchw_to_tokens(image.to(self.device, torch.float32), ld=64)
        conv1, H2, W2 = self.conv1(x, n, H, W)                                   # 1/2, 64 ch
        y, H4, W4 = ops.pool2d(conv1, n, H2, W2, 64, 3, 2, pad=1)                # 1/4
        layer1 = None
        h, w = H4, W4
        for li, blocks in enumerate(self.layers):
            for blk in blocks:
                y, h, w = blk(y, n, h, w)
            if li == 0:
                layer1 = y                                                       # 1/4, 256 ch
        H8, W8 = h, w
        # --- decoder input: cat(img_enc 256, sparse_enc 16) -> 272 channels in a 320-wide (zero padded) buffer
        cat = torch.zeros((n * H8 * W8, 320), dtype=torch.float16, device=self.device)
        self.conv5(y, n, H8, W8, out=cat[:, :256])
        # --- sparse encoder (shallownet.py:12-21)
        s = ops.nchw_to_tokens(sparse.to(self.device, torch.float32), ld=64)
        s, hs, ws = self.s0(s, n, H, W)                                          # 5x5 s2 -> 1/2, 16 ch (ld 64)
        s, hs, ws = ops.pool2d(s, n, hs, ws, 64, 2, 2)                           # max 2 -> 1/4
        s, _, _ = self.s4(s, n, hs, ws)
        s, hs, ws = ops.pool2d(s, n, hs, ws, 64, 2, 2, mode="avg")               # avg 2 -> 1/8
        assert (hs, ws) == (H8, W8)
        ops.copy2d(s[:, :16], cat[:, 256:272])
        # --- decoder (decoder.py:188-213)
        cat512 = torch.empty((n * H8 * W8, 512), dtype=torch.float16, device=self.device)
        for bi, (name, k) in enumerate((("decoder1", 1), ("decoder2", 2), ("decoder4", 4), ("decoder8", 8))):
            t, h, w = (cat, H8, W8) if k == 1 else ops.pool2d(cat, n, H8, W8, 320, k, k)
            for c in self.dec[name]:
                t, _, _ = c(t, n, h, w)
            dst = cat512[:, bi * 128:(bi + 1) * 128]
            if k == 1:
                ops.copy2d(t, dst)
            else:
                ops.resize_bilinear_ac(t, n, h, w, 128, H8, W8, out=dst)
        f8, _, _ = self.fusion8(cat512, n, H8, W8)                               # 256
        c4 = torch.empty((n * H4 * W4, 384), dtype=torch.float16, device=self.device)
        ops.resize_bilinear_ac(f8, n, H8, W8, 256, H4, W4, out=c4[:, :256])
        self.skipconv4(layer1, n, H4, W4, out=c4[:, 256:384])
        f4, _, _ = self.fusion4(c4, n, H4, W4)                                   # 128
        c2 = torch.zeros((n * H2 * W2, 192), dtype=torch.float16, device=self.device)   # 128 + 32 (+ 32 zero pad)
        ops.resize_bilinear_ac(f4, n, H4, W4, 128, H2, W2, out=c2[:, :128])
        self.skipconv2(conv1, n, H2, W2, out=c2[:, 128:160])
        f2, _, _ = self.fusion2(c2, n, H2, W2)                                   # 64
        logits, _, _ = self.head(f2, n, H2, W2)                                  # 198 (+2 pad rows) in a 256-wide buffer
        return logits, H2, W2


class CMP_demo:
    """``CMP_demo`` (…_norefine.py:25-62) without yaml / checkpoint-directory plumbing: takes the CMP state_dict."""

    def __init__(self, state_dict, device="cuda"):
        self.model = CMP(state_dict, device)
        self.device = device

    @torch.no_grad()
    def run(self, image, sparse, mask):
        dtype = image.dtype
        n, _, H, W = image.shape
        img = image.to(self.device, torch.float32) * 2 - 1
        sp = torch.cat([sparse, mask], dim=1).to(self.device, torch.float32)
        logits, h, w = self.model.forward(img, sp)
        flow = ops.flow_expectation(logits, n, h, w, NBINS, FMAX)                # Fuser.convert_flow
        if h != H:
            flow = ops.resize_bilinear_ac_f32(flow, H, W)
        return flow.to(dtype)


def get_cmp_flow(cmp, frames, sparse_optical_flow, mask, brush_mask=None):      # run_gradio.py:236-258
    b, t, c, h, w = frames.shape
    flow = cmp.run(frames.flatten(0, 1), sparse_optical_flow.flatten(0, 1), mask.flatten(0, 1))
    if brush_mask is not None:
        bm = (torch.as_tensor(brush_mask) / 255.).to(flow.device, dtype=flow.dtype)
        flow = flow * bm.unsqueeze(0).unsqueeze(0)
    return flow.reshape(b, t, 2, h, w)


def get_flow(cmp, pixel_values_384, sparse_optical_flow_384, mask_384, height, width, motion_brush_mask=None):
    """run_gradio.py:261-277 (the working size is the input's; the reference fixes it at 384)."""
    fb, fl = pixel_values_384.shape[:2]
    hs, ws = pixel_values_384.shape[-2:]
    flow = get_cmp_flow(cmp, pixel_values_384[:, 0:1].repeat(1, fl, 1, 1, 1), sparse_optical_flow_384, mask_384,
                        motion_brush_mask)
    if height != hs or width != ws:
        f = ops.resize_nearest_f32(flow.reshape(fb * fl * 2, hs, ws).float().contiguous(), height, width)
        flow = f.reshape(fb, fl, 2, height, width)
        flow[:, :, 0] *= width / ws
        flow[:, :, 1] *= height / hs
    return flow
