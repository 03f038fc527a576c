"""Oracle (test infrastructure): Euler discrete scheduler restated from
MOFA-Video-Traj/utils/scheduling_euler_discrete_karras_fix.py
(``__init__`` :178-236, ``init_noise_sigma`` :249-255, ``scale_model_input`` :264-288,
``set_timesteps`` :290-350, ``_sigma_to_t`` :352-374, ``_convert_to_karras`` :376-399,
``step`` :418-528).  Defaults = SVD-XT scheduler_config.json.
"""
import numpy as np
import torch

SVD_XT_SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                        beta_schedule="scaled_linear", prediction_type="v_prediction",
                        interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002, sigma_max=700.0,
                        timestep_spacing="leading", timestep_type="continuous", steps_offset=1)


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, **kw):
        cfg = dict(SVD_XT_SCHEDULER)
        cfg.update(kw)
        self.config = cfg
        n = cfg["num_train_timesteps"]
        if cfg["beta_schedule"] == "linear":
            self.betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
        elif cfg["beta_schedule"] == "scaled_linear":
            self.betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(cfg["beta_schedule"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.use_karras_sigmas = cfg["use_karras_sigmas"]
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()[::-1].copy()
        if self.use_karras_sigmas:
            sigmas = self._convert_to_karras(sigmas, n)
        sigmas = torch.from_numpy(sigmas).to(torch.float32)
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self.timesteps = None
        self.num_inference_steps = None
        self._step_index = None

    @property
    def init_noise_sigma(self):
        max_sigma = self.sigmas.max()
        if self.config["timestep_spacing"] in ["linspace", "trailing"]:
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def _convert_to_karras(self, in_sigmas, num_inference_steps):
        sigma_min = self.config.get("sigma_min")
        sigma_max = self.config.get("sigma_max")
        sigma_min = sigma_min if sigma_min is not None else in_sigmas[-1].item()
        sigma_max = sigma_max if sigma_max is not None else in_sigmas[0].item()
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        min_inv_rho = sigma_min ** (1 / rho)
        max_inv_rho = sigma_max ** (1 / rho)
        return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho

    def _sigma_to_t(self, sigma, log_sigmas):
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = log_sigmas[low_idx], log_sigmas[high_idx]
        w = np.clip((low - log_sigma) / (low - high), 0, 1)
        return ((1 - w) * low_idx + w * high_idx).reshape(sigma.shape)

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.config["num_train_timesteps"]
        sp = self.config["timestep_spacing"]
        if sp == "linspace":
            timesteps = np.linspace(0, n - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif sp == "leading":
            step_ratio = n // num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
            timesteps += self.config["steps_offset"]
        elif sp == "trailing":
            step_ratio = n / num_inference_steps
            timesteps = (np.arange(n, 0, -step_ratio)).round().copy().astype(np.float32) - 1
        else:
            raise ValueError(sp)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        log_sigmas = np.log(sigmas)
        if self.config["interpolation_type"] == "linear":
            sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        else:
            raise ValueError(self.config["interpolation_type"])
        if self.use_karras_sigmas:
            sigmas = self._convert_to_karras(sigmas, num_inference_steps)
            timesteps = np.array([self._sigma_to_t(s, log_sigmas) for s in sigmas])
        sigmas = torch.from_numpy(sigmas).to(torch.float32)
        if self.config["timestep_type"] == "continuous" and self.config["prediction_type"] == "v_prediction":
            self.timesteps = torch.Tensor([0.25 * s.log() for s in sigmas])
        else:
            self.timesteps = torch.from_numpy(timesteps.astype(np.float32))
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self._step_index = None

    def _init_step_index(self, timestep):
        idx = (self.timesteps == timestep).nonzero()
        self._step_index = (idx[1] if len(idx) > 1 else idx[0]).item()

    def scale_model_input(self, sample, timestep):
        if self.step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self.step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample):
        """s_churn = 0 path of :418-528 (gamma = 0; the reference's unused randn draw has no effect)."""
        if self.step_index is None:
            self._init_step_index(timestep)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self.step_index]
        sigma_hat = sigma
        pt = self.config["prediction_type"]
        if pt == "v_prediction":
            pred_original_sample = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        elif pt == "epsilon":
            pred_original_sample = sample - sigma_hat * model_output
        else:
            raise ValueError(pt)
        derivative = (sample - pred_original_sample) / sigma_hat
        dt = self.sigmas[self.step_index + 1] - sigma_hat
        prev_sample = (sample + derivative * dt).to(model_output.dtype)
        self._step_index += 1
        return prev_sample
