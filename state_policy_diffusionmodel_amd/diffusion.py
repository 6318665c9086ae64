"""Host-side mirror of the reference's sampler objects for the hot path:
``Diffusion_DDPM`` (``/root/reference/models/diffusion_ddpm.py:22-88, 216-277, 283-348``) and
``Diffusion_DDIM`` (``models/diffusion_ddim.py:19-74``).

Same constructor keywords, same attributes a caller touches (``noise_scheduler``,
``noise_steps``, ``noise_estimator``, ``obs_horizon``, ``pred_horizon``, ``inpaint_horizon``,
``prediction_dim``, ``vision_encoder``), same ``sample(batch, option)`` signature and return
types -- so ``generate.py``'s idiom

    model.noise_scheduler = DDIMScheduler(num_train_timesteps=100, ...)   # generate.py:28-34
    model.noise_steps = 100                                               # generate.py:35
    history = model.sample(batch=obs, option='sample_history')            # generate.py:73-76

works unchanged.  What runs underneath is libspdm_hip.so (no torch compute in the loop).
Training, validation plots and the Lightning plumbing are out of scope.

Explicit, non-breaking extensions: ``sample(..., x_T=, noise=, batched=, seed=)`` for
fixed-noise parity runs and for B > 1 independent trajectories (the reference hard-wires
B = 1 by taking ``obs_cond[0]``, ``models/diffusion_ddpm.py:246``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from .engine import SpdmEngine
from .schedulers import DDIMScheduler, DDPMScheduler, _LinearBetaScheduler
from .weights import random_state_dict, state_dict_to_numpy


def _as_spec(sched) -> _LinearBetaScheduler:
    """Accept our own scheduler objects, or a diffusers-like one (duck-typed on class name and
    ``config``), as ``generate.py`` may assign either."""
    if isinstance(sched, _LinearBetaScheduler):
        return sched
    name = type(sched).__name__
    cfg = getattr(sched, "config", None)
    T = getattr(cfg, "num_train_timesteps", None) if cfg is not None else None
    if T is None:
        raise TypeError(f"cannot interpret noise_scheduler of type {name}")
    kw = dict(num_train_timesteps=int(T), beta_start=float(getattr(cfg, "beta_start", 1e-4)),
              beta_end=float(getattr(cfg, "beta_end", 0.02)),
              beta_schedule=getattr(cfg, "beta_schedule", "linear"),
              clip_sample=bool(getattr(cfg, "clip_sample", False)),
              prediction_type=getattr(cfg, "prediction_type", "epsilon"))
    if "DDIM" in name:
        return DDIMScheduler(**kw)
    if "DDPM" in name:
        return DDPMScheduler(**kw)
    raise TypeError(f"unsupported scheduler class {name} (DDPM / DDIM only)")


class NoiseEstimator:
    """``self.noise_estimator`` of the reference (``UNet_Film(...)`` built at
    models/diffusion_ddpm.py:76-82): holds the weights; calling it evaluates the HIP U-Net."""

    def __init__(self, owner: "Diffusion_DDPM", state_dict):
        self._owner = owner
        self._sd = state_dict_to_numpy(state_dict)

    def state_dict(self):
        return {k: torch.from_numpy(v) for k, v in self._sd.items()}

    def __call__(self, x: torch.Tensor, t: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        eng = self._owner._engine_for(x.shape[0], x.shape[-2], x.shape[-1])
        return eng.unet_forward(x, t, y)


class Diffusion_DDPM:
    def __init__(self, noise_steps: int = 1000, obs_horizon: int = 10, pred_horizon: int = 10,
                 observation_dim: int = 2, prediction_dim: int = 2, learning_rate: float = 1e-4,
                 model: str = "UNet", vision_encoder: Optional[Callable] = None,
                 noise_scheduler_type: str = "linear", inpaint_horizon: int = 10, step_size: int = 1,
                 *, state_dict=None, weight_seed: int = 0, device: int = 0, max_batch: int = 1,
                 vision_encoder_state_dict=None):
        # --- Diffusion params (models/diffusion_ddpm.py:42-48)
        self.noise_steps = noise_steps
        self.obs_horizon = obs_horizon
        self.pred_horizon = pred_horizon
        self.observation_dim = observation_dim
        self.prediction_dim = prediction_dim
        self.inpaint_horizon = inpaint_horizon
        self.lr = learning_rate
        # --- architecture switch (:54-62); the concat-conditioned 'UNet' is a different network
        if model == "UNet_Film":
            self.attention = True
        elif model == "UNet_FilmnoAttention":
            self.attention = False
        else:
            raise NotImplementedError("only model='UNet_Film' / 'UNet_FilmnoAttention' are on the MI355X path "
                                      "(simple_Unet.py is out of scope, SURVEY.md section 2)")
        # --- scheduler (:65-70); beta schedule is hard-coded 'linear' there, noise_scheduler_type unused
        self.noise_scheduler = DDPMScheduler(num_train_timesteps=self.noise_steps, beta_schedule="linear",
                                             clip_sample=False, prediction_type="epsilon")
        self.cond_dim = observation_dim * obs_horizon
        if state_dict is None:   # random-init, like constructing the reference module without a checkpoint
            state_dict = random_state_dict(self.cond_dim, seed=weight_seed, attention=self.attention)
        self.noise_estimator = NoiseEstimator(self, state_dict)
        # the reference loads a private autoencoder checkpoint here (:84-88).  Given that encoder's tensors
        # (vision_encoder_state_dict: keys 0.weight .. 7.bias, as in the diffusion checkpoint's 'vision_encoder.*'), the
        # front end runs in libspdm_hip.so (vision.VisionEncoder, built on first use); any other callable
        # (N,3,96,96) -> (N,128) can be plugged in instead, or the batch may carry 'image_features'
        self.vision_encoder = vision_encoder
        self._vision_sd = vision_encoder_state_dict
        self.device = torch.device("cuda", device)
        self._device_index = device
        self._max_batch = max_batch
        self._engine: Optional[SpdmEngine] = None
        self._engine_key = None

    # ------------------------------------------------------------------------------------------
    _HPARAM_KEYS = ("noise_steps", "obs_horizon", "pred_horizon", "observation_dim", "prediction_dim", "learning_rate",
                    "model", "noise_scheduler_type", "inpaint_horizon", "step_size")

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, hparams_file=None, map_location=None, **kwargs):
        """Lightning's ``LightningModule.load_from_checkpoint(ckpt, hparams_file=yaml)`` as generate.py:25,27 and
        run_predictions.py call it: constructor arguments from ``hparams.yaml`` (``save_hyperparameters()`` of
        models/diffusion_ddpm.py:37), U-Net tensors from the checkpoint's ``noise_estimator.*`` entries, the observation
        encoder's from its ``vision_encoder.*`` entries when present (vision.VisionEncoder)."""
        from .weights import check_state_dict, fetch_hyperparams_from_yaml, load_checkpoint_state_dict
        hp = dict(fetch_hyperparams_from_yaml(hparams_file)) if hparams_file else {}
        ctor = {k: hp[k] for k in cls._HPARAM_KEYS if k in hp}
        if isinstance(hp.get("vision_encoder"), str) or hp.get("vision_encoder") is None:
            pass                                   # a name (e.g. 'resnet18') in the yaml is not a callable: ignored
        ctor.update(kwargs)
        sd, other = load_checkpoint_state_dict(str(checkpoint_path))
        attention = ctor.get("model", "UNet") != "UNet_FilmnoAttention"
        cond_dim = int(ctor.get("observation_dim", 2)) * int(ctor.get("obs_horizon", 10))
        check_state_dict(sd, cond_dim, attention=attention)
        if "vision_encoder_state_dict" not in ctor and any(k.startswith("vision_encoder.") for k in other):
            from .vision import encoder_state_dict_from
            from .weights import safe_load_state_dict
            ctor["vision_encoder_state_dict"] = encoder_state_dict_from(safe_load_state_dict(str(checkpoint_path)))
        return cls(state_dict=sd, **ctor)

    # Lightning look-alikes used by callers (generate.py:36)
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------------------------------
    def _engine_for(self, batch: int, H: int, D: int, pin: bool = False) -> SpdmEngine:
        spec = _as_spec(self.noise_scheduler)
        T = max(int(spec.config.num_train_timesteps), int(self.noise_steps))
        key = (H, D, T, batch if pin else 0)       # (a pinned engine is tied to ONE global batch)
        if self._engine is None or self._engine_key != key or batch > self._engine.max_batch:
            if self._engine is not None:
                self._engine.close()
            self._engine = SpdmEngine(H, D, self.cond_dim, max_batch=batch if pin else max(batch, self._max_batch),
                                      device=self._device_index, attention=self.attention,
                                      num_train_timesteps=T, pin_geometry=pin)
            self._engine.load_state_dict(self.noise_estimator._sd)
            self._engine_key = key
        return self._engine

    def add_constraints(self, x_t: torch.Tensor, x_inpaint: torch.Tensor) -> torch.Tensor:
        """models/diffusion_ddpm.py:216-219 (in place, broadcast over the batch)."""
        x_t[:, :, :self.inpaint_horizon, :] = x_inpaint
        return x_t

    # ==================== Sampling (models/diffusion_ddpm.py:223-277) ====================
    def sample(self, batch: Dict[str, torch.Tensor], option: Optional[str] = None, *,
               x_T: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
               batched: bool = False, seed: Optional[int] = None, sample_offset: int = 0, every: int = 1,
               sharded: Optional[bool] = None, group=None, shard_exact: bool = False):
        """``option``: None -> x_0 (B,1,H,D); 'sample_history' -> list of the N+1 iterates (the reference's form);
        'sample_history_stream' -> a generator of ``(i, x_i)`` host tensors handed out while the loop runs
        (``every``: stride in steps; SpdmEngine.sample_stream).

        ``seed``: key of the device noise stream that replaces the ``torch.randn`` diffusers' DDPM ``step`` draws from
        the global generator on every step of every call.  None (default) draws a FRESH 62-bit seed per call from
        torch's global generator -- so, as with the reference, two calls give different trajectories and
        ``torch.manual_seed`` makes a run reproducible.  Ignored when ``noise`` is supplied.

        ``sharded`` (with ``batched=True``): every rank of the process group passes the SAME global batch; each runs its
        contiguous slice of the trajectories on its own GPU (no communication inside the loop) and one all-gather -- RCCL
        over xGMI on the "nccl" backend -- returns all B trajectories on every rank, rank-major, independent of the rank
        count (distributed.ShardedSampler).  None (default): shard iff a process group with more than one rank is
        initialised.  ``x_T`` and ``seed`` left at None are drawn on rank 0 and broadcast.  A shard's trajectories agree
        with the single-GPU run to fp32 rounding (<= 1e-5 over a loop: a small shard selects other kernels for the coarse
        levels); ``shard_exact=True`` pins every rank's kernel selection to that of the GLOBAL batch instead -- bit-identical
        to the single-GPU run, at the price of small-batch kernels not being used on small shards."""
        from .distributed import ShardedSampler, shard_bounds, world_and_rank
        world, rank = world_and_rank(group)
        if sharded is None:
            sharded = batched and world > 1
        if sharded and not batched:
            raise ValueError("sharded=True needs batched=True (the reference's B = 1 form has nothing to shard)")
        if sharded and option == "sample_history_stream":
            raise ValueError("option='sample_history_stream' hands out this process's iterates: not available with sharded=True")
        if seed is None:
            seed_t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
            if sharded and world > 1:
                seed_t = self._broadcast0(seed_t, group)
            seed = int(seed_t.item())
        for key, tensor in batch.items():
            batch[key] = tensor.to(self.device)
        obs_cond = self.prepare_obs_cond_vectors(batch)                       # (B, obs_h, obs_dim)
        inpaint = self.prepare_inpaint_vectors(batch)                         # (B, inp_h, pred_dim)
        if not batched:                                                       # reference: B forced to 1
            obs_cond, inpaint = obs_cond[0:1], inpaint[0:1]
        obs_cond = obs_cond.unsqueeze(1)                                      # (B,1,obs_h,obs_dim)
        inpaint = inpaint.unsqueeze(1)                                        # (B,1,inp_h,pred_dim)
        B = obs_cond.shape[0]
        H, D = self.pred_horizon + self.inpaint_horizon, self.prediction_dim
        if x_T is None:
            x_T = torch.rand(B, 1, H, D, device=self.device)                  # uniform, :252
            if sharded and world > 1:
                x_T = self._broadcast0(x_T, group)
        spec = _as_spec(self.noise_scheduler)
        spec.set_timesteps(self.noise_steps)                                  # :257/:268
        s0, s1 = shard_bounds(B, rank, world) if sharded else (0, B)
        eng = self._engine_for(B if (sharded and shard_exact) else s1 - s0, H, D, pin=bool(sharded and shard_exact))
        eng.set_scheduler(spec)
        ip = inpaint if self.inpaint_horizon > 0 else None
        if option == "sample_history_stream":
            return eng.sample_stream(obs_cond, x_T, noise=noise, inpaint=ip, seed=seed, sample_offset=sample_offset, every=every)
        want_hist = (option == "sample_history")
        if sharded:
            if sample_offset:
                raise ValueError("sample_offset is the shard's own bookkeeping when sharded=True")
            res = ShardedSampler(eng, group).sample(obs_cond, x_T, noise=noise, inpaint=ip, seed=seed, history=want_hist)
        else:
            res = eng.sample(obs_cond, x_T, noise=noise, inpaint=ip, seed=seed, sample_offset=sample_offset, history=want_hist)
        if want_hist:
            _, hist = res
            return [hist[i] for i in range(hist.shape[0])]                    # list of N+1 (B,1,H,D), :256-265
        return res

    def _broadcast0(self, t: torch.Tensor, group=None) -> torch.Tensor:
        """rank 0's value of ``t`` on every rank (through the device for a backend that only moves device tensors)."""
        import torch.distributed as dist
        dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        buf = t.to(dev).contiguous()
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return buf.to(t.device)

    # ==================== Helper functions (models/diffusion_ddpm.py:283-348) ====================
    def prepare_observation_batch(self, batch):
        out = {}
        for k in ("image", "position", "action", "velocity", "image_features"):
            if k in batch:
                out[k] = batch[k][:, :self.obs_horizon].to(self.device).float()
        return out

    def prepare_obs_cond_vectors(self, observation_batch):
        if "obs_cond" in observation_batch:
            return observation_batch["obs_cond"].float()
        if "image_features" in observation_batch:
            feats = observation_batch["image_features"]
        else:
            if self.vision_encoder is None and self._vision_sd is not None:
                from .vision import VisionEncoder
                self.vision_encoder = VisionEncoder(self._vision_sd, device=self._device_index)
            if self.vision_encoder is None:
                raise RuntimeError("batch has raw images but neither vision_encoder nor vision_encoder_state_dict was "
                                   "supplied (the reference's autoencoder checkpoint is not part of the repo)")
            img = observation_batch["image"]
            with torch.no_grad():
                enc = self.vision_encoder(img.flatten(end_dim=1))
            feats = enc.reshape(*img.shape[:2], -1)
        return torch.cat([observation_batch["position"], observation_batch["action"],
                          observation_batch["velocity"], feats], dim=-1)

    def prepare_inpaint_vectors(self, observation_batch):
        if "inpaint" in observation_batch:
            return observation_batch["inpaint"].float()
        if self.inpaint_horizon == 0:
            B = next(iter(observation_batch.values())).shape[0]
            return torch.zeros(B, 0, self.prediction_dim, device=self.device)
        pos = observation_batch["position"][:, -self.inpaint_horizon:, :]
        act = observation_batch["action"][:, -self.inpaint_horizon:, :]
        return torch.cat([pos, act], dim=-1)

    def prepare_prediction_batch(self, batch):
        """models/diffusion_ddpm.py:300-315: everything after the observed window, ``batch[k][:, self.obs_horizon:]``
        (the dataset windows are obs_horizon + pred_horizon long, so this is the last pred_horizon entries)."""
        out = {}
        for k in ("image", "position", "action", "velocity", "image_features"):
            if k in batch:
                out[k] = batch[k][:, self.obs_horizon:].to(self.device).float()
        return out

    def prepare_prediction_vectors(self, prediction_batch):
        """models/diffusion_ddpm.py:332-338: x_0 = cat(position, action)."""
        return torch.cat([prediction_batch["position"], prediction_batch["action"]], dim=-1)

    # ==================== Validation (models/diffusion_ddpm.py:175-214) ====================
    def validate(self, batch, *, x_T: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """The reference's validation sampler: the first trajectory of the batch through the full loop; returns
        ``(x_0 (1,1,H,D), observation_batch, inpaint_vector (1,1,inp_h,D))``."""
        observation_batch = self.prepare_observation_batch(batch)
        inpaint_vector = self.prepare_inpaint_vectors(observation_batch)[0:1].unsqueeze(1)
        x_0 = self.sample(dict(observation_batch), x_T=x_T, noise=noise)
        return x_0, observation_batch, inpaint_vector

    # ==================== Training, forward half (models/diffusion_ddpm.py:140-172) ====================
    def training_step(self, batch, batch_idx: int = 0, *, t: Optional[torch.Tensor] = None,
                      noise: Optional[torch.Tensor] = None, return_parts: bool = False):
        """The forward computation of the reference's ``training_step``: noising of the target window at a per-sample
        timestep (``add_noise``), in-painting of the observed rows, ONE U-Net evaluation with ``t`` of shape (B,), MSE
        against the noise.  The U-Net runs on the HIP path (``spdm_unet_forward`` with per-sample t); there is no
        backward pass here -- the returned loss carries no graph (training itself is outside this path).
        ``t`` / ``noise`` may be passed for reproducibility (the reference draws them with torch.randint / randn_like)."""
        observation_batch = self.prepare_observation_batch(batch)
        prediction_batch = self.prepare_prediction_batch(batch)
        obs_cond = self.prepare_obs_cond_vectors(observation_batch).unsqueeze(1)            # (B,1,obs_h,obs_dim)
        x_0 = self.prepare_prediction_vectors(prediction_batch).unsqueeze(1)               # (B,1,pred_h,pred_dim)
        x_0_inpaint = self.prepare_inpaint_vectors(observation_batch).unsqueeze(1)         # (B,1,inp_h,pred_dim)
        B = x_0.shape[0]
        if t is None:
            t = torch.randint(0, self.noise_steps, (B,), device=self.device)
        t = t.to(self.device).long()
        prediction_vector = torch.cat([x_0_inpaint, x_0], dim=2)                           # concat in time
        if noise is None:
            noise = torch.randn_like(prediction_vector)
        noise = noise.to(self.device).float()
        x_noisy = _as_spec(self.noise_scheduler).add_noise(prediction_vector, noise, t)
        x_noisy = self.add_constraints(x_noisy, x_0_inpaint)
        noise_estimated = self.noise_estimator(x_noisy, t, obs_cond)
        loss = torch.mean((noise - noise_estimated) ** 2)                                  # nn.MSELoss, :49
        return (loss, noise_estimated, x_noisy) if return_parts else loss

    def validation_step(self, batch, batch_idx: int = 0, **kw):
        return self.training_step(batch, batch_idx, **kw)


class Diffusion_DDIM(Diffusion_DDPM):
    """models/diffusion_ddim.py:19-74: a byte-identical copy of the DDPM loop; it becomes DDIM only
    because the caller swaps ``noise_scheduler`` (generate.py:28-35).  Nothing to override here."""
    pass


def load_model(model_name: str, checkpoint_path=None, hparams_path=None, num_of_ddim_steps: int = 100, *,
               state_dict=None, **hparams):
    """generate.py:23-37: build the sampler from a checkpoint + hparams.yaml (same positional signature as the
    reference) -- or from an in-memory ``state_dict`` / random init when no checkpoint is given -- and, for DDIM,
    overwrite scheduler and noise_steps exactly as the reference's loader does."""
    if model_name not in ("DDPM", "DDIM"):
        raise ValueError("model_name must be 'DDPM' or 'DDIM'")
    cls = Diffusion_DDPM if model_name == "DDPM" else Diffusion_DDIM
    if checkpoint_path is not None and not isinstance(checkpoint_path, (str, bytes)) and not hasattr(checkpoint_path, "__fspath__"):
        state_dict, checkpoint_path = checkpoint_path, None        # load_model(name, state_dict) of earlier callers
    if checkpoint_path is not None:
        model = cls.load_from_checkpoint(checkpoint_path, hparams_file=hparams_path, **hparams)
    else:
        model = cls(state_dict=state_dict, **hparams)
    if model_name == "DDIM":
        model.noise_scheduler = DDIMScheduler(num_train_timesteps=num_of_ddim_steps, beta_schedule="linear",
                                              clip_sample=False, prediction_type="epsilon")
        model.noise_steps = num_of_ddim_steps
    model.eval()
    return model
