"""mofa_video_amd -- MI355X-native (gfx950) MOFA-Video denoising hot path.

Host side mirrors the reference's Python surface (pipeline / adapter / UNet / softsplat call
signatures and diffusers ``state_dict`` keys); all arithmetic runs in hand-written HIP kernels in
``libmofa_hip.so`` behind the C ABI of ``include/mofa_hip.h``.  There is no CPU or eager-PyTorch
fallback: without the built library every op raises.
"""
__version__ = "0.1.0"
