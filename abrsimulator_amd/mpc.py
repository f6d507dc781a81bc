"""BatchedMPCController: the reference's MPCBitrateController (mpc.py:20-186)
for many players at once.  next_bitrate() is one launch of the K3 kernel
(csrc/abr_env.hip: mpc_select_kernel) through the C ABI.
"""
import ctypes as C
import os

import torch

from . import _lib
from .datamodel import MPD, QOEMetric


class BatchedMPCController:
    """Mirror of MPCBitrateController(player, bitrate_utility, horizon) (mpc.py:52).

    `player` follows the reference's player protocol (mpc.py:56-57,166,184):
      get_mpd() -> MPD with per-chunk .bitrates/.sizes, .chunk_length, .max_buffer
      get_qoe_metric() -> QOEMetric (.variance_weight/.rebuffer_weight/.startup_weight)
      get_next_chunk_info() -> object with tensors over lanes:
          chunk_number i32[N], previous_bitrate i32[N], buffer_level f64[N], and the
          history summary hist_n f64[N] / hist_sum_inv f64[N] standing for
          previous_bandwidths (len and sum(1/x) in list order); optional `mask` u8[N].
    As in the reference the bitrate_utility argument is ignored (mpc.py:58): the
    utility is the identity (mpc.py:95-97).  horizon defaults to 3 (mpc.py:59).
    Reference quirks reproduced: D9 (next_bitrate grows the player's history by
    `horizon` predictions), D10 (3-argument max, unclamped rebuffer term), D11
    (lookahead buffer uses the current chunk's sizes).  clip_horizon=True defines
    the behaviour where the reference raises IndexError (D12).
    """

    METHODS = {"harmonic": 0, "expsmoothing": 1}
    UTILITIES = {"identity": 0, "log": 1}

    def __init__(self, player=None, bitrate_utility=None, horizon=None, clip_horizon=True,
                 device="cuda", *, method="harmonic", utility="identity"):
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.horizon = 3 if horizon is None else int(horizon)
        self.clip_horizon = bool(clip_horizon)
        # the reference's alternative predictor / utility (mpc.py:72-79, :99-102): PARITY UNPINNED,
        # see include/abr_env.h: abr_mpc_options.  method="expsmoothing" needs the throughput
        # history itself: chunk-info fields `previous_bandwidths` f64[T, N] (entry t of lane i)
        # and `history_length` i32[N]  (EnvPlayer provides both).
        if method not in self.METHODS or utility not in self.UTILITIES:
            raise ValueError("method is 'harmonic' or 'expsmoothing'; utility is 'identity' or 'log'")
        self.method, self.utility = method, utility
        # run the predictor as its own kernel (False: everything in one kernel; same results)
        self.use_scratch = os.environ.get("ABR_MPC_SINGLE_KERNEL") != "1"
        self._scratch = None
        self._bound = None             # (key, config, options): rebuilt only when N / tables / weights change
        self.player = None
        self._tables_for = None
        if player is not None:
            self.player = player
            self.mpd = player.get_mpd()
            self.qoe = player.get_qoe_metric()
        self.last_flat = None
        self.last_J = None

    # the two refresh hooks the reference declares (mpc.py:61-67; broken there: no self)
    def update_mpd(self):
        self.mpd = self.player.get_mpd()
        self._tables_for = None
        self._bound = None

    def update_qoe(self):
        self.qoe = self.player.get_qoe_metric()
        self._bound = None

    def default_bitrate_utility(self, bitrate):
        return bitrate

    def _tables(self):
        mpd = self.mpd
        if self._tables_for is not mpd:
            chunks = mpd.chunk_list()
            L = float(mpd.chunk_length)
            br = [[float(b) for b in c.bitrates] for c in chunks]
            sz = [[float(s) for s in (c.sizes if c.sizes is not None else
                                      [b * L for b in c.bitrates])] for c in chunks]
            self.br = torch.tensor(br, dtype=torch.float64, device=self.device).contiguous()
            self.sz = torch.tensor(sz, dtype=torch.float64, device=self.device).contiguous()
            self._tables_for = mpd
        return self.br, self.sz

    def config(self):
        br, _ = self._tables()
        c = _lib.MpcConfig()
        c.n_rates, c.horizon, c.video_length = br.shape[1], int(self.horizon), br.shape[0]
        c.clip_horizon = int(self.clip_horizon)
        c.chunk_length, c.max_buffer = float(self.mpd.chunk_length), float(self.mpd.max_buffer)
        c.variance_weight = float(self.qoe.variance_weight)
        c.rebuffer_weight = float(self.qoe.rebuffer_weight)
        c.startup_weight = float(self.qoe.startup_weight)
        return c

    def _bind_key(self, n_lanes):
        """Everything the bound config / options depend on: a change of any of it -- also an in-place change of the MPD's
        chunk_length or max_buffer, which config() reads -- rebinds on the next select."""
        return (int(n_lanes), self._tables_for, self.horizon, self.clip_horizon, self.method, self.utility,
                float(self.mpd.chunk_length), float(self.mpd.max_buffer),
                float(self.qoe.variance_weight), float(self.qoe.rebuffer_weight), float(self.qoe.startup_weight))

    def next_bitrate(self, want_details=False):
        """mpc.py:181-186, batched: returns int32 [N] bitrate indices."""
        ci = self.player.get_next_chunk_info()
        br, sz = self._tables()
        N = int(ci.chunk_number.numel())
        action = torch.empty(N, dtype=torch.int32, device=self.device)
        flat = torch.empty(N, dtype=torch.int32, device=self.device) if want_details else None
        J = torch.empty(N, dtype=torch.float64, device=self.device) if want_details else None
        # `mask`: lanes with a zero byte are skipped; `done`: an environment's ABR_DONE_* bytes as they
        # are (lanes with a NON-zero byte are skipped) -- no host-side tensor work to turn one into the other
        mask, mask_is_done = getattr(ci, "mask", None), 0
        if mask is None and getattr(ci, "done", None) is not None:
            mask, mask_is_done = ci.done, 1
        for t, dt in ((ci.chunk_number, torch.int32), (ci.previous_bitrate, torch.int32),
                      (ci.buffer_level, torch.float64), (ci.hist_n, torch.float64),
                      (ci.hist_sum_inv, torch.float64)):
            if t.dtype != dt or t.device.type != "cuda":
                raise TypeError(f"chunk-info tensors must be {dt} on the GPU")
        # the config / options structs and the predictor's scratch are bound once per (lane count, tables, weights,
        # method): a select is then the two kernel launches and nothing else on the host (BoundOut's counterpart)
        key = self._bind_key(N)
        if self._bound is None or self._bound[0] != key:
            cfg = self.config()
            opt = _lib.MpcOptions()
            opt.predictor, opt.utility = self.METHODS[self.method], self.UTILITIES[self.utility]
            if self.use_scratch:
                # scratch for the predictor pre-kernel (include/abr_env.h: abr_mpc_options.scratch_dev)
                need = C.c_size_t()
                _lib.check(self.lib.abr_mpc_scratch_bytes(C.byref(cfg), N, C.byref(need)))
                if self._scratch is None or self._scratch.numel() < need.value:
                    self._scratch = torch.empty(need.value, dtype=torch.uint8, device=self.device)
                opt.scratch_dev, opt.scratch_bytes = self._scratch.data_ptr(), self._scratch.numel()
            self._bound = (key, cfg, opt)
        _, cfg, opt = self._bound
        opt.mask_is_done = mask_is_done
        if self.method == "expsmoothing":
            hist, hlen = ci.previous_bandwidths, ci.history_length
            if hist.dtype != torch.float64 or hist.dim() != 2 or hist.shape[1] != N or hist.stride(1) != 1:
                raise TypeError("previous_bandwidths must be float64 [T, N] with unit lane stride")
            if hlen.dtype != torch.int32 or int(hlen.max()) > hist.shape[0]:
                raise TypeError("history_length must be int32 [N], at most T")
            opt.hist_dev, opt.hist_stride = hist.data_ptr(), hist.stride(0)
            opt.hist_len_dev = hlen.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.abr_mpc_select_opt(
                C.byref(cfg), C.byref(opt),
                _lib.ptr(ci.chunk_number), _lib.ptr(ci.previous_bitrate),
                _lib.ptr(ci.buffer_level), _lib.ptr(ci.hist_n), _lib.ptr(ci.hist_sum_inv),
                _lib.ptr(br), _lib.ptr(sz), _lib.ptr(mask), _lib.ptr(action), _lib.ptr(flat),
                _lib.ptr(J), N, _lib.current_stream(self.device)))
        self.last_flat, self.last_J = flat, J
        return action

    def objective_grid(self, chunk, prev_bitrate, buffer_level, predicted_bandwidths):
        """objective() (mpc.py:120-162) over the whole brute grid of ONE player, float64 [B^H]."""
        br, sz = self._tables()
        cfg = self.config()
        pred = torch.as_tensor(predicted_bandwidths, dtype=torch.float64, device=self.device).contiguous()
        out = torch.empty(cfg.n_rates ** cfg.horizon, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.abr_mpc_objective_grid(
                C.byref(cfg), int(chunk), int(prev_bitrate), float(buffer_level), _lib.ptr(pred),
                _lib.ptr(br), _lib.ptr(sz), _lib.ptr(out), _lib.current_stream(self.device)))
        return out


class EnvPlayer:
    """Adapter closing the gap the reference leaves open (D5/D6): exposes a
    BatchedABREnv through the player protocol mpc.py expects, zero-copy -- the
    chunk-info tensors ARE the environment's state, so the predictor's history
    mutation (D9) lands in the environment's previous_bandwidths summary exactly
    as the reference's shared list would."""

    class _Info:
        pass

    def __init__(self, env, mpd: MPD = None, qoe: QOEMetric = None):
        self.env = env
        self._mpd = mpd if mpd is not None else env.mpd
        self._qoe = qoe if qoe is not None else env.qoe_metric
        (self.chunk_id, self.last_bitrate, self.buffer_level, self.hist_n, self.hist_sum_inv,
         self.done) = env.mpc_inputs()

    def get_mpd(self):
        return self._mpd

    def get_qoe_metric(self):
        return self._qoe

    def get_next_chunk_info(self):
        ci = EnvPlayer._Info()
        ci.chunk_number, ci.previous_bitrate = self.chunk_id, self.last_bitrate
        ci.buffer_level, ci.hist_n, ci.hist_sum_inv = self.buffer_level, self.hist_n, self.hist_sum_inv
        ci.done = self.done            # the kernel reads the done bits themselves (abr_mpc_options.mask_is_done)
        # the list itself, for predictors that need more than its harmonic summary (f4):
        # previous_bandwidths[t, i] for t < chunk_id[i]
        ci.previous_bandwidths = self.env.history()[1]
        ci.history_length = self.chunk_id
        return ci
