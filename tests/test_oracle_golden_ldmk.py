"""CPU tests: the oracle's landmark adapter (ForegroundMatting hourglass, landmark embedding, zero-outs), the Hybrid
dual-adapter step with mask blend and the Keypoint window loop against fixtures produced by running the reference's
own MOFA-Video-Hybrid / MOFA-Video-Keypoint code (tests/golden/make_golden_ldmk.py)."""
import os

import pytest
import torch

from helpers import LDMK_CN, LDMK_UNET, rel_l2, synthetic_inputs, synthetic_landmarks
from mofa_video_amd import schema

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def models():
    from oracle.controlnet import FlowControlNet
    from oracle.ldmk import LandmarkFlowControlNet
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel
    face = LandmarkFlowControlNet(**LDMK_CN)
    drag = FlowControlNet(**LDMK_CN)
    un = UNetSpatioTemporalConditionControlNetModel(**LDMK_UNET)
    face.load_state_dict({k: t.float() for k, t in schema.synthetic_state_dict(schema.ldmk_controlnet_schema(LDMK_CN), seed=11).items()})
    drag.load_state_dict({k: t.float() for k, t in schema.synthetic_state_dict(schema.controlnet_schema(LDMK_CN), seed=12).items()})
    un.load_state_dict({k: t.float() for k, t in schema.synthetic_state_dict(schema.unet_schema(LDMK_UNET), seed=10).items()})
    return face.eval(), drag.eval(), un.eval()


def test_ldmk_inventory_and_forward(models):
    face, drag, un = models
    G = torch.load(os.path.join(GD, "reference_golden_hybrid.pt"), weights_only=False)
    assert schema.ldmk_controlnet_schema() == G["inventory"]["ldmk_full"]           # full-size key/shape inventory
    assert {k: tuple(v.shape) for k, v in face.state_dict().items()} == G["inventory"]["ldmk_reduced"]
    a = G["ldmk_forward"]
    T, H, W = a["T"], a["H"], a["W"]
    inp = synthetic_inputs(T, H, W, cross_dim=LDMK_CN["cross_attention_dim"], seed=43)
    lm = synthetic_landmarks(T, H, W, seed=44)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    with torch.no_grad():
        dr, mr, _, om = face(a["xin"], torch.tensor(a["timestep"]), inp["image_embeddings"], ids,
                             controlnet_cond=torch.cat([inp["cond"]] * 2), controlnet_flow=torch.cat([inp["flow"]] * 2),
                             landmarks=torch.cat([lm] * 2), return_dict=False, conditioning_scale=a["conditioning_scale"])
    for i, (x, y) in enumerate(zip(list(dr) + [mr], list(a["down"]) + [a["mid"]])):
        assert rel_l2(x, y) < 2e-4, (i, rel_l2(x, y))
    for x, y in zip(om, a["occlusion_masks"]):
        assert tuple(x.shape) == tuple(y.shape) and rel_l2(x, y) < 1e-4


def test_hybrid_pipeline(models):
    from oracle.pipeline import denoise_hybrid
    from oracle.scheduler import EulerDiscreteScheduler
    face, drag, un = models
    p = torch.load(os.path.join(GD, "reference_golden_hybrid.pt"), weights_only=False)["hybrid_pipeline"]
    il = torch.cat([torch.zeros_like(p["image_latents"]), p["image_latents"]])
    emb = torch.cat([torch.zeros_like(p["image_embeddings"]), p["image_embeddings"]])
    lat = denoise_hybrid(un, face, drag, EulerDiscreteScheduler(), p["latents_in"], il, emb, p["cond"], p["flow"],
                         p["landmarks"], p["drag_flow"], p["mask"], num_inference_steps=p["steps"],
                         ctrl_scale_traj=p["ctrl_scale_traj"], ctrl_scale_ldmk=p["ctrl_scale_ldmk"])
    assert rel_l2(lat, p["final_latents"]) < 2e-4, rel_l2(lat, p["final_latents"])


def test_keypoint_window_loop(models):
    from oracle.pipeline import denoise_keypoint_loop, window_views
    from oracle.scheduler import EulerDiscreteScheduler
    face, drag, un = models
    p = torch.load(os.path.join(GD, "reference_golden_keypoint.pt"), weights_only=False)["keypoint_pipeline"]
    assert window_views(25, 25, 12) == [(1, 25), (1, 25)]            # config 3: the same window twice
    assert window_views(49, 25, 12) == [(1, 25), (13, 37), (25, 49), (25, 49)]
    assert window_views(p["N"], p["window_size"], p["stride"]) == [(1, 4), (3, 6), (3, 6)]
    il = torch.cat([torch.zeros_like(p["image_latents"]), p["image_latents"]])
    emb = torch.cat([torch.zeros_like(p["image_embeddings"]), p["image_embeddings"]])
    lat = denoise_keypoint_loop(un, face, EulerDiscreteScheduler(), p["latents_in"], il, emb, p["cond"], p["flow"],
                                p["landmarks"], window_size=p["window_size"], stride=p["stride"],
                                num_inference_steps=p["steps"])
    assert rel_l2(lat, p["final_latents"]) < 2e-4, rel_l2(lat, p["final_latents"])
