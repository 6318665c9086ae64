"""SpdmEngine -- thin Python owner of one ``spdm_handle`` (include/spdm.h).

PyTorch-ROCm tensors are used for interop only (device memory, streams): every
call hands ``data_ptr()`` / the current HIP stream to the C ABI; all compute is
in libspdm_hip.so.  No CPU path exists here.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from .weights import pack_state_dict, unet_film_param_spec


def reference_time_table(T: int, channels: int = 256) -> torch.Tensor:
    """pos_encoding(t) for t = 0..T-1 with torch's own fp32 ops in the reference's order
    (models/Unet_FiLmLayer.py:266-274 applied to ``t.unsqueeze(-1).float()``, :281)."""
    t = torch.arange(T, dtype=torch.int64).unsqueeze(-1).type(torch.float)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2) / channels))
    a = torch.sin(t.repeat(1, channels // 2) * inv_freq)
    b = torch.cos(t.repeat(1, channels // 2) * inv_freq)
    return torch.cat([a, b], dim=-1).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class SpdmEngine:
    def __init__(self, horizon: int, state_dim: int, cond_dim: int, max_batch: int, device: int = 0,
                 attention: bool = True, time_dim: int = 256, num_train_timesteps: int = 1000,
                 debug: bool = False, exact_fp32: bool = False, pin_geometry: bool = False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("SpdmEngine needs a visible MI355X (HIP device); there is no CPU fallback")
        self.device = torch.device("cuda", device)
        self.horizon, self.state_dim, self.cond_dim = int(horizon), int(state_dim), int(cond_dim)
        self.max_batch, self.attention, self.time_dim = int(max_batch), bool(attention), int(time_dim)
        self.num_train_timesteps = int(num_train_timesteps)
        cfg = _lib.SpdmConfig(self.horizon, self.state_dim, self.cond_dim, self.time_dim, int(self.attention),
                              self.max_batch, device, self.num_train_timesteps,
                              (_lib.SPDM_FLAG_DEBUG_KEEP if debug else 0) | (_lib.SPDM_FLAG_EXACT_FP32 if exact_fp32 else 0))
        h = ctypes.c_void_p()
        _lib.check(self.lib.spdm_create(ctypes.byref(cfg), ctypes.byref(h)), "spdm_create")
        self._h = h
        if pin_geometry:
            # kernel selection (tile sizes, split-K, the small-grid kernel, fused sources) normally follows the batch of the call;
            # pinned, every call runs the kernels a batch of max_batch would -- so a rank that holds a shard of a larger batch
            # (max_batch = the GLOBAL batch) reproduces the single-GPU trajectories bit for bit (multi-GPU)
            _lib.check(self.lib.spdm_set_switch(self._h, b"SPDM_PIN_GEOMETRY", 1), "spdm_set_switch")
        self._keep = []           # tensors the C side reads asynchronously during a session
        self.n_steps = 0
        self.kind = None
        tab = reference_time_table(self.num_train_timesteps, self.time_dim).numpy()
        _lib.check(self.lib.spdm_set_time_table(self._h, tab.ctypes.data_as(ctypes.c_void_p),
                                                self.num_train_timesteps), "spdm_set_time_table")

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.spdm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def split_precision(self) -> bool:
        """True: contractions on the split-fp16 MFMA path (default); False: exact fp32 MFMA."""
        return bool(self.lib.spdm_uses_split_precision(self._h))

    @property
    def device_bytes(self) -> int:
        return int(self.lib.spdm_device_bytes(self._h))

    @property
    def demoted_tensors(self) -> int:
        """Weight tensors outside the split format's range (|w| >= 511), kept on the exact fp32 kernels."""
        return int(self.lib.spdm_demoted_tensors(self._h))

    @property
    def graph_captures(self) -> int:
        """Step graphs captured so far: stays at 1 across sample() calls of one shape, whatever tensors they pass."""
        return int(self.lib.spdm_graph_captures(self._h))

    def set_switch(self, name: str, on: bool = True) -> None:
        """Flip one kernel-selection switch (``SPDM_NO_GRAPH``, ``SPDM_NO_WIDE``, ...) on this handle.  The
        environment is read once, at construction; this is the test / tuning hook for a live engine."""
        _lib.check(self.lib.spdm_set_switch(self._h, name.encode(), int(bool(on))), "spdm_set_switch")

    def nonfinite(self) -> bool:
        """True if the last sampling loop / U-Net evaluation produced a non-finite value (synchronises the stream).
        On the split-precision path this is how an activation beyond its range (|x| > 4094) shows."""
        flag = ctypes.c_int32(0)
        _lib.check(self.lib.spdm_nonfinite(self._h, ctypes.byref(flag), self._stream()), "spdm_nonfinite")
        return bool(flag.value)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t: Optional[torch.Tensor], shape=None, name="tensor") -> Optional[torch.Tensor]:
        if t is None:
            return None
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            t = t.reshape(shape)
        return t

    # -- setup --------------------------------------------------------------------------------
    def load_state_dict(self, sd) -> None:
        """state_dict of the reference noise predictor (names as ``UNet_Film.state_dict()``;
        a Lightning checkpoint's ``noise_estimator.`` prefix is stripped)."""
        blob, idx = pack_state_dict(sd)
        want = set(unet_film_param_spec(self.cond_dim, self.time_dim, self.attention).keys())
        have = {e.name.decode() for e in idx}
        missing = sorted(want - have)
        if missing:
            raise KeyError(f"state_dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        _lib.check(self.lib.spdm_load_weights(self._h, blob.ctypes.data_as(ctypes.c_void_p), blob.size, idx,
                                              len(idx)), "spdm_load_weights")

    def set_scheduler(self, sched) -> None:
        """Install the tables of a host scheduler object (schedulers.py) for the loop."""
        ts = np.ascontiguousarray(np.asarray(sched.timesteps, dtype=np.int64).astype(np.int32))
        coef = np.ascontiguousarray(sched.coefficient_table(), dtype=np.float32)
        assert coef.shape == (ts.size, 6)
        _lib.check(self.lib.spdm_set_schedule_tables(self._h, int(sched.kind), int(ts.size),
                                                     ts.ctypes.data_as(ctypes.c_void_p),
                                                     coef.ctypes.data_as(ctypes.c_void_p)), "spdm_set_schedule_tables")
        self.n_steps, self.kind = int(ts.size), int(sched.kind)

    def set_builtin_schedule(self, kind: int, num_train_timesteps: int, num_inference_steps: int,
                             beta_start: float = 1e-4, beta_end: float = 0.02) -> None:
        _lib.check(self.lib.spdm_set_schedule(self._h, kind, num_train_timesteps, num_inference_steps,
                                              beta_start, beta_end), "spdm_set_schedule")
        self.n_steps, self.kind = int(num_inference_steps), int(kind)

    # -- the noise predictor ------------------------------------------------------------------
    def unet_forward(self, x: torch.Tensor, t, cond: Optional[torch.Tensor]) -> torch.Tensor:
        """eps = noise_estimator(x, t, cond); x (B,1,H,D), t (1,)|(B,) ints, cond (B,1,obs_h,obs_dim)."""
        B = x.shape[0]
        xs = self._dev(x, (B, self.horizon, self.state_dim))
        cs = self._dev(cond, (B, self.cond_dim)) if cond is not None else None
        tt = np.ascontiguousarray(torch.as_tensor(t).reshape(-1).cpu().numpy().astype(np.int32))
        eps = torch.empty((B, 1, self.horizon, self.state_dim), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.spdm_unet_forward(self._h, B, _ptr(xs), tt.ctypes.data_as(ctypes.c_void_p), int(tt.size),
                                              _ptr(cs), _ptr(eps), self._stream()), "spdm_unet_forward")
        return eps

    # -- the sampling loop --------------------------------------------------------------------
    def sample_begin(self, cond, x_T, noise=None, inpaint=None, seed: int = 0, sample_offset: int = 0,
                     history: bool = False) -> Optional[torch.Tensor]:
        if self.n_steps <= 0:
            raise RuntimeError("no scheduler installed (set_scheduler)")
        B = x_T.shape[0]
        H, D = self.horizon, self.state_dim
        xs = self._dev(x_T, (B, H, D))
        cs = self._dev(cond, (B, self.cond_dim)) if cond is not None else None
        ns = self._dev(noise, (self.n_steps, B, H, D)) if noise is not None else None
        inp_h, per_sample, ip = 0, 0, None
        if inpaint is not None:
            ip = self._dev(inpaint)
            ip = ip.reshape(-1, ip.shape[-2], ip.shape[-1])
            inp_h = ip.shape[1]
            if ip.shape[2] != D or ip.shape[0] not in (1, B):
                raise ValueError(f"inpaint must be (1|B,1,inp_h,{D}), got {tuple(inpaint.shape)}")
            per_sample = int(ip.shape[0] == B and B > 1)
        hist = torch.empty((self.n_steps + 1, B, 1, H, D), device=self.device, dtype=torch.float32) if history else None
        self._keep = [xs, cs, ns, ip, hist]
        self._B = B
        _lib.check(self.lib.spdm_sample_begin(self._h, B, _ptr(cs), _ptr(ip), inp_h, per_sample, _ptr(xs), _ptr(ns),
                                              ctypes.c_uint64(seed), ctypes.c_uint64(sample_offset), _ptr(hist),
                                              self._stream()), "spdm_sample_begin")
        return hist

    def sample_run(self, step_begin: int, step_end: int) -> None:
        _lib.check(self.lib.spdm_sample_run(self._h, step_begin, step_end, self._stream()), "spdm_sample_run")

    def sample_result(self) -> torch.Tensor:
        out = torch.empty((self._B, 1, self.horizon, self.state_dim), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.spdm_sample_result(self._h, _ptr(out), self._stream()), "spdm_sample_result")
        return out

    def sample(self, cond, x_T, noise=None, inpaint=None, seed: int = 0, sample_offset: int = 0,
               history: bool = False, check_finite: bool = True):
        """x_0 (B,1,H,D); with history=True also the (n_steps+1,B,1,H,D) stack x_T..x_0.  Raises if an iterate left
        the finite range (the split-precision contractions overflow beyond |activation| 4094: re-create the engine
        with exact_fp32=True for such a model)."""
        hist = self.sample_begin(cond, x_T, noise, inpaint, seed, sample_offset, history)
        self.sample_run(0, self.n_steps)
        out = self.sample_result()
        if check_finite and self.nonfinite():
            raise FloatingPointError("non-finite iterate in the sampling loop" + (
                ": an activation left the split-fp16 range (|x| < 4094); use exact_fp32=True" if self.split_precision else ""))
        return (out, hist) if history else out

    def sample_stream(self, cond, x_T, noise=None, inpaint=None, seed: int = 0, sample_offset: int = 0, every: int = 1):
        """Generator over the running loop: yields ``(i, x_i)`` -- the iterate after ``i`` denoise steps as a HOST tensor
        (B,1,H,D) -- for i = 0, every, 2*every, ..., n_steps, WHILE the device keeps stepping: chunk k+1 is enqueued before
        chunk k's snapshot is handed out, and snapshots travel on a side stream into pinned memory.  This is the streaming
        form of ``option='sample_history'`` (models/diffusion_ddpm.py:256-265 builds the whole list first; its consumer,
        utils/plot_utils.py:199-277, draws one frame per iterate)."""
        if every < 1:
            raise ValueError("every must be >= 1")
        self.sample_begin(cond, x_T, noise, inpaint, seed, sample_offset, history=False)
        main = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        pending = None                                     # (step index, pinned host tensor, copy-done event)

        def snapshot(i):
            dev_copy = self.sample_result()                # device-side copy, ordered behind step i on the loop's stream
            ready = torch.cuda.Event()
            ready.record(main)
            host = torch.empty(dev_copy.shape, dtype=dev_copy.dtype, pin_memory=True)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                host.copy_(dev_copy, non_blocking=True)
                dev_copy.record_stream(side)
                done = torch.cuda.Event()
                done.record(side)
            return i, host, done

        pending = snapshot(0)
        i = 0
        while i < self.n_steps:
            j = min(self.n_steps, i + every)
            self.sample_run(i, j)                          # asynchronous: the device works on [i, j) ...
            nxt = snapshot(j)
            pi, ph, pd = pending                           # ... while the caller consumes iterate i
            pd.synchronize()
            yield pi, ph
            pending, i = nxt, j
        pi, ph, pd = pending
        pd.synchronize()
        yield pi, ph

    # -- introspection ------------------------------------------------------------------------
    def debug_tensor(self, name: str) -> torch.Tensor:
        """Named intermediate of the last unet_forward as NCHW (engine created with debug=True)."""
        shape = (ctypes.c_int32 * 4)()
        _lib.check(self.lib.spdm_debug_tensor(self._h, name.encode(), ctypes.c_void_p(0), 0, ctypes.byref(shape)),
                   "spdm_debug_tensor")
        B, H, W, C = (int(v) for v in shape)
        buf = torch.empty((B, H, W, C), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.spdm_debug_tensor(self._h, name.encode(), _ptr(buf), buf.numel(), ctypes.byref(shape)),
                   "spdm_debug_tensor")
        return buf.permute(0, 3, 1, 2).contiguous()

    def profile(self, on) -> None:
        """True / False: per-launch HIP events on / off; "prepare": create the events now, instrument nothing yet."""
        mode = 2 if on == "prepare" else int(bool(on))
        _lib.check(self.lib.spdm_profile_enable(self._h, mode), "spdm_profile_enable")

    def profile_read(self):
        n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(self.lib.spdm_profile_read(self._h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)),
                   "spdm_profile_read")
        return int(n.value), float(ms.value), float(fl.value)
