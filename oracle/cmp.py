"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the CMP sparse-to-dense motion encoder at inference
(SURVEY N1), with the reference's module / parameter names so that one state_dict loads into both:

* ``ResNet`` (Bottleneck [3,4,6,3], layer3 / layer4 de-strided with dilation 2 / 4, 1x1 ``conv5`` head):
  Traj/models/cmp/models/backbone/resnet.py:49-166
* ``ShallowNet`` (shallownet8x): Traj/models/cmp/models/modules/shallownet.py:4-41
* ``MotionDecoderSkipLayer``: Traj/models/cmp/models/modules/decoder.py:96-213
* ``CMP`` (module): Traj/models/cmp/models/modules/cmp.py:6-37, configuration
  Traj/models/cmp/experiments/semiauto_annot/resnet50_vip+mpii_liteflow/config.yaml
* ``Fuser.convert_flow`` (99-bin softmax expectation): Traj/models/cmp/utils/visualize_utils.py:6-19
* ``CMPDemo.run``: Traj/models/svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine.py:51-62
* ``get_cmp_flow`` / ``get_flow``: Traj/run_gradio.py:236-277

Pinned by tests/golden/reference_golden_cmp.pt: the reference's own classes, imported in place and loaded with the same
seeded state_dict (tests/golden/make_golden_cmp.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

CMP_PARAMS = dict(image_encoder="resnet50", sparse_encoder="shallownet8x", flow_decoder="MotionDecoderSkipLayer",
                  skip_layer=True, img_enc_dim=256, sparse_enc_dim=16, output_dim=198, nbins=99, fmax=50)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(out + residual)


class ResNet(nn.Module):
    def __init__(self, output_dim, layers=(3, 4, 6, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.conv5 = nn.Conv2d(2048, output_dim, kernel_size=1)
        for layer, d in ((self.layer3, 2), (self.layer4, 4)):      # resnet.py:118-129: de-stride, dilate
            for n, m in layer.named_modules():
                if "conv2" in n:
                    m.dilation, m.padding, m.stride = (d, d), (d, d), (1, 1)
                elif "downsample.0" in n:
                    m.stride = (1, 1)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, img, ret_feat=False):
        conv1 = self.relu(self.bn1(self.conv1(img)))               # 1/2
        layer1 = self.layer1(self.maxpool(conv1))                  # 1/4
        out = self.conv5(self.layer4(self.layer3(self.layer2(layer1))))   # 1/8
        return (out, [img, conv1, layer1]) if ret_feat else out


class ShallowNet(nn.Module):
    def __init__(self, input_dim=4, output_dim=16, stride=(2, 2, 2)):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(input_dim, 16, kernel_size=5, stride=stride[0], padding=2), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=stride[1], stride=stride[1]),
            nn.Conv2d(16, output_dim, kernel_size=3, padding=1), nn.BatchNorm2d(output_dim), nn.ReLU(inplace=True),
            nn.AvgPool2d(kernel_size=stride[2], stride=stride[2]))

    def forward(self, x):
        return self.features(x)


def _cbr(cin, cout):
    return [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class MotionDecoderSkipLayer(nn.Module):
    def __init__(self, input_dim=512, output_dim=2):
        super().__init__()
        self.decoder1 = nn.Sequential(*_cbr(input_dim, 128), *_cbr(128, 128), *_cbr(128, 128))
        self.decoder2 = nn.Sequential(nn.MaxPool2d(2, 2), *_cbr(input_dim, 128), *_cbr(128, 128), *_cbr(128, 128))
        self.decoder4 = nn.Sequential(nn.MaxPool2d(4, 4), *_cbr(input_dim, 128), *_cbr(128, 128), *_cbr(128, 128))
        self.decoder8 = nn.Sequential(nn.MaxPool2d(8, 8), *_cbr(input_dim, 128), *_cbr(128, 128), *_cbr(128, 128))
        self.fusion8 = nn.Sequential(*_cbr(512, 256))
        self.skipconv4 = nn.Sequential(*_cbr(256, 128))
        self.fusion4 = nn.Sequential(*_cbr(256 + 128, 128))
        self.skipconv2 = nn.Sequential(*_cbr(64, 32))
        self.fusion2 = nn.Sequential(*_cbr(128 + 32, 64))
        self.head = nn.Conv2d(64, output_dim, kernel_size=1, padding=0)

    def forward(self, x, skip_feat):
        layer1, layer2, layer4 = skip_feat                         # (img, conv1 at 1/2, layer1 at 1/4)

        def up(t, ref):
            return F.interpolate(t, size=(ref.size(2), ref.size(3)), mode="bilinear", align_corners=True)
        x1 = self.decoder1(x)
        cat = torch.cat([x1, up(self.decoder2(x), x1), up(self.decoder4(x), x1), up(self.decoder8(x), x1)], dim=1)
        f8 = self.fusion8(cat)
        f4 = self.fusion4(torch.cat([up(f8, layer4), self.skipconv4(layer4)], dim=1))
        f2 = self.fusion2(torch.cat([up(f4, layer2), self.skipconv2(layer2)], dim=1))
        return self.head(f2)


class CMP(nn.Module):
    def __init__(self, params=CMP_PARAMS):
        super().__init__()
        self.image_encoder = ResNet(params["img_enc_dim"])
        self.flow_encoder = ShallowNet(output_dim=params["sparse_enc_dim"], stride=(2, 2, 2))
        self.flow_decoder = MotionDecoderSkipLayer(input_dim=params["img_enc_dim"] + params["sparse_enc_dim"],
                                                   output_dim=params["output_dim"])

    def forward(self, image, sparse):
        sparse_enc = self.flow_encoder(sparse)
        img_enc, skip_feat = self.image_encoder(image, ret_feat=True)
        return self.flow_decoder(torch.cat((img_enc, sparse_enc), dim=1), skip_feat)


class Fuser:
    def __init__(self, nbins, fmax):
        self.nbins, self.fmax = nbins, fmax
        self.step = 2 * fmax / float(nbins)
        self.mesh = torch.arange(nbins).view(1, -1, 1, 1).float() * self.step - fmax + self.step / 2

    def convert_flow(self, flow_prob):
        px = F.softmax(flow_prob[:, :self.nbins], dim=1) * self.mesh
        py = F.softmax(flow_prob[:, self.nbins:], dim=1) * self.mesh
        return torch.cat([px.sum(dim=1, keepdim=True), py.sum(dim=1, keepdim=True)], dim=1)


class CMPDemo(nn.Module):
    """CMP_demo without the checkpoint / yaml loading: ``model`` is the CMP module in eval mode."""

    def __init__(self, params=CMP_PARAMS):
        super().__init__()
        self.model = CMP(params).eval()
        self.fuser = Fuser(params["nbins"], params["fmax"])

    @torch.no_grad()
    def run(self, image, sparse, mask):                            # ..._norefine.py:51-62
        dtype = image.dtype
        image = image * 2 - 1
        out = self.model(image.float(), torch.cat([sparse, mask], dim=1).float())
        flow = self.fuser.convert_flow(out)
        if flow.shape[2] != image.shape[2]:
            flow = F.interpolate(flow, size=image.shape[2:4], mode="bilinear", align_corners=True)
        return flow.to(dtype)


def get_cmp_flow(cmp, frames, sparse_optical_flow, mask, brush_mask=None):      # run_gradio.py:236-258
    b, t, c, h, w = frames.shape
    flow = cmp.run(frames.flatten(0, 1), sparse_optical_flow.flatten(0, 1), mask.flatten(0, 1))
    if brush_mask is not None:
        bm = (torch.as_tensor(brush_mask) / 255.).to(flow.device, dtype=flow.dtype)
        flow = flow * bm.unsqueeze(0).unsqueeze(0)
    return flow.reshape(b, t, 2, h, w)


def get_flow(cmp, pixel_values_384, sparse_optical_flow_384, mask_384, height, width, motion_brush_mask=None):
    """run_gradio.py:261-277: first frame repeated, CMP at the working size, nearest resize + per-axis rescale."""
    fb, fl = pixel_values_384.shape[:2]
    hs, ws = pixel_values_384.shape[-2:]
    flow = get_cmp_flow(cmp, pixel_values_384[:, 0:1].repeat(1, fl, 1, 1, 1), sparse_optical_flow_384, mask_384,
                        motion_brush_mask)
    if height != hs or width != ws:
        scales = [height / hs, width / ws]
        flow = F.interpolate(flow.flatten(0, 1), (height, width), mode="nearest").reshape(fb, fl, 2, height, width)
        flow[:, :, 0] *= scales[1]
        flow[:, :, 1] *= scales[0]
    return flow
