"""Host mirror of the reference Euler scheduler's *schedule* logic
(MOFA-Video-Traj/utils/scheduling_euler_discrete_karras_fix.py: ``set_timesteps`` :290-350,
``_convert_to_karras`` :376-399, ``init_noise_sigma`` :249-255, step-index bookkeeping :401-416).

Only scalars live here (sigma table, continuous timesteps); the tensor math of
``scale_model_input`` (:264-288) and ``step`` (:418-528) runs in libmofa_hip.so
(``mofa_prepare_model_input`` / ``mofa_cfg_euler_step``), fused with the CFG combine.
"""
import math

import numpy as np

SVD_XT_SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                        beta_schedule="scaled_linear", prediction_type="v_prediction",
                        interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002, sigma_max=700.0,
                        timestep_spacing="leading", timestep_type="continuous", steps_offset=1)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, **kwargs):
        cfg = dict(SVD_XT_SCHEDULER)
        cfg.update(kwargs)
        self.config = _Cfg(cfg)
        if cfg["prediction_type"] != "v_prediction":
            raise NotImplementedError("the SVD/MOFA path is v-prediction")
        n = cfg["num_train_timesteps"]
        if cfg["beta_schedule"] == "scaled_linear":
            betas = np.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=np.float32) ** 2
        elif cfg["beta_schedule"] == "linear":
            betas = np.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=np.float32)
        else:
            raise NotImplementedError(cfg["beta_schedule"])
        acp = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self._train_sigmas = (((1 - acp) / acp) ** 0.5).astype(np.float32)
        self.sigmas = None
        self.timesteps = None
        self.num_inference_steps = None
        self._step_index = None
        self.set_timesteps(n)

    def _karras(self, in_sigmas, num):
        smin = self.config.sigma_min if self.config.get("sigma_min") is not None else float(in_sigmas[-1])
        smax = self.config.sigma_max if self.config.get("sigma_max") is not None else float(in_sigmas[0])
        rho = 7.0
        ramp = np.linspace(0, 1, num)
        return (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho

    def set_timesteps(self, num_inference_steps, device=None):
        cfg = self.config
        n = cfg.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        sp = cfg.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy().astype(np.float32)
            ts += cfg.steps_offset
        elif sp == "trailing":
            ts = np.arange(n, 0, -n / num_inference_steps).round().copy().astype(np.float32) - 1
        else:
            raise ValueError(sp)
        sig = np.interp(ts, np.arange(0, n), self._train_sigmas)
        if cfg.use_karras_sigmas:
            sig = self._karras(sig, num_inference_steps)
        sig = sig.astype(np.float32)
        if cfg.timestep_type == "continuous":
            self.timesteps = np.array([0.25 * math.log(float(s)) for s in sig], dtype=np.float32)
        else:
            self.timesteps = ts
        self.sigmas = np.concatenate([sig, np.zeros(1, dtype=np.float32)])
        self._step_index = None

    @property
    def init_noise_sigma(self):
        m = float(self.sigmas.max())
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return m
        return (m * m + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def step_table(self):
        """fp32 [num_inference_steps, 8] rows {sigma, sigma_next, timestep, timestep, 1 / sqrt(sigma^2 + 1), 0, 0, 0}: everything
        a denoise step takes from the scheduler (scale_model_input :284-285, step :504-520, the network's timestep), uploaded
        once per clip (include/mofa_hip.h, MOFA_STEP_SCALARS) -- float32 arithmetic as the host-scalar entry points do it"""
        n = len(self.timesteps)
        tab = np.zeros((n, 8), dtype=np.float32)
        sig = self.sigmas.astype(np.float32)
        tab[:, 0], tab[:, 1] = sig[:n], sig[1:n + 1]
        tab[:, 2] = tab[:, 3] = self.timesteps
        tab[:, 4] = np.float32(1.0) / np.sqrt(sig[:n] * sig[:n] + np.float32(1.0))
        return tab

    def sigma_pair(self, i):
        return float(self.sigmas[i]), float(self.sigmas[i + 1])
