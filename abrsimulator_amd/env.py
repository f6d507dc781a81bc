"""BatchedABREnv: the reference's Simulator.run() tick loop (Simulator.py:93-210)
turned inside out into reset()/step(actions) over many independent lanes.

Host Python only holds PyTorch-ROCm tensors and calls the C ABI
(include/abr_env.h); all simulation runs in the HIP kernels of csrc/abr_env.hip.
"""
import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import F64_DIM, F64_ROWS, OBS_DIM, OBS_ROWS
from .datamodel import MPD, NetworkInfo, QOEMetric


def pack_traces(traces: Sequence, device):
    """Ragged list of bandwidth traces -> (flat f64, offsets i64, lengths i32) on `device`."""
    ts = [torch.as_tensor(t, dtype=torch.float64).reshape(-1) for t in traces]
    if not ts or any(t.numel() == 0 for t in ts):
        raise ValueError("every trace needs at least one bandwidth sample")
    lens = torch.tensor([t.numel() for t in ts], dtype=torch.int32)
    off = torch.zeros(len(ts), dtype=torch.int64)
    off[1:] = torch.cumsum(lens[:-1].to(torch.int64), 0)
    flat = torch.cat(ts)
    # a NaN/inf/negative bandwidth can never complete a download: the lane would crawl to
    # max_ticks one real addition at a time (bounded, but pointlessly slow)
    if not bool(torch.isfinite(flat).all()) or bool((flat < 0).any()):
        raise ValueError("bandwidth traces must be finite and non-negative")
    return flat.to(device), off.to(device), lens.to(device)


class BoundOut(dict):
    """An output dict of step_random / step_script whose device pointers have been looked up once
    (BatchedABREnv.bind_out): a launch per call is then a ctypes call and nothing else.  The tensors
    must not be replaced afterwards."""
    ptrs = None


class BatchedABREnv:
    """n_lanes independent players, one per GPU thread.

    mpd / qoe_metric / network_info carry the reference's field names
    (datamodel.py).  reset() returns the observation at each lane's first
    get_next_bitrate call site (Simulator.py:155); step(actions) supplies that
    call's return value and returns (obs, reward, done) at the next one.
    Observations are float32 [OBS_DIM, n_lanes] (rows: _lib.OBS_ROWS); the exact
    float64 state is available through observe_f64() and state_view().
    """

    def __init__(self, mpd: MPD, qoe_metric: QOEMetric, network_info: NetworkInfo, n_lanes: int,
                 device="cuda", speed=1.0, auto_reset: bool = False, max_ticks: int = 0,
                 lane_id_base: int = 0, impl: str = "auto", library: Optional[str] = None):
        # `library`: a diagnostic build of the same ABI (tools/diag/csrc/Makefile -> tools/diag/lib), by path or file
        # name.  The product library holds only what `auto` can select plus the jump / split / tick cross-checks;
        # impl="async" / "ring3" (pipelines that were measured slower, kept for the parity tests and the records)
        # need such a library and are refused (ABR_E_UNSUPPORTED) by the product.
        self.lib = _lib.lib(library)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("BatchedABREnv runs on a ROCm device only (no CPU path exists)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._dev_index = self.device.index
        self.n_lanes = int(n_lanes)
        self.mpd, self.qoe_metric, self.network_info = mpd, qoe_metric, network_info
        # one ladder for the whole video (what run() indexes, Simulator.py:82,156) or, for an MPD
        # whose chunks differ (set_mpd's one-ladder-per-line file), a per-chunk table
        self.br_table = None
        if mpd.uniform():
            ladder = mpd.ladder()
        else:
            table = mpd.bitrate_table()
            if len(table) != int(mpd.video_length) or any(len(r) != len(table[0]) for r in table):
                raise ValueError("a per-chunk MPD needs video_length ladders of equal length")
            ladder = table[0]
            self.br_table = torch.tensor(table, dtype=torch.float64, device=self.device).contiguous()
            if not bool((self.br_table > 0).all()):
                raise ValueError("bitrates must be > 0")
        cfg = _lib.EnvConfig()
        cfg.n_rates, cfg.video_length = len(ladder), int(mpd.video_length)
        cfg.chunk_length, cfg.max_buffer = float(mpd.chunk_length), float(mpd.max_buffer)
        cfg.start_up_length, cfg.interval = float(mpd.start_up_length), float(network_info.interval)
        cfg.rebuffer_weight = float(qoe_metric.rebuffer_weight)
        cfg.variance_weight = float(qoe_metric.variance_weight)
        cfg.startup_weight = float(qoe_metric.startup_weight)
        cfg.latency_weight = float(getattr(qoe_metric, "latency_weight", 0.0))
        self.lane_speeds = None
        if torch.is_tensor(speed) or hasattr(speed, "__len__"):
            # per-lane play speeds: [N] = one constant speed per lane;
            # [rows, N] = a speed controller's answers, one row per played chunk
            # (Simulator.py:176-177; the last row repeats)
            ls = torch.as_tensor(speed, dtype=torch.float64)
            if ls.dim() == 1:
                ls = ls.reshape(1, -1)
            if (ls.dim() != 2 or ls.shape[1] != self.n_lanes or ls.shape[0] < 1
                    or not bool((ls > 0).all()) or not bool(torch.isfinite(ls).all())):
                raise ValueError("per-lane speeds must be [n_lanes] or [rows, n_lanes] finite values > 0")
            self.lane_speeds = ls.to(self.device).contiguous()
            speed = 1.0
        cfg.speed = float(speed)
        if len(ladder) > _lib.MAX_RATES:
            raise ValueError(f"at most {_lib.MAX_RATES} bitrates")
        for i, b in enumerate(ladder):
            cfg.ladder[i] = b
        cfg.max_ticks, cfg.auto_reset = int(max_ticks), int(bool(auto_reset))
        self.cfg = cfg
        self.n_rates, self.video_length = cfg.n_rates, cfg.video_length

        bw = network_info.bandwidths
        if torch.is_tensor(bw):
            bw = [bw] if bw.dim() == 1 else list(bw)
        elif len(bw) > 0 and not hasattr(bw[0], "__len__"):
            bw = [bw]              # one trace given as a flat list of floats
        with torch.cuda.device(self.device):
            self.traces, self.trace_off, self.trace_len = pack_traces(bw, self.device)
            self.n_traces = int(self.trace_len.numel())
            nbytes = C.c_size_t()
            self._check(self.lib.abr_env_workspace_bytes(C.byref(cfg), self.n_lanes, C.byref(nbytes)))
            self.workspace = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            assert self.workspace.data_ptr() % 256 == 0
            h = C.c_void_p()
            self._check(self.lib.abr_env_create(
                C.byref(cfg), _lib.ptr(self.traces), _lib.ptr(self.trace_off),
                _lib.ptr(self.trace_len), self.n_traces, self.n_lanes, _lib.ptr(self.workspace),
                nbytes.value, self._stream(), C.byref(h)))
        self._h = h
        # the layout tag abr_env_create wrote into the workspace's last bytes, kept aside: what a state to be loaded must carry
        self._tag = self.workspace[-self.TAG_BYTES:].clone()
        if lane_id_base:
            self._check(self.lib.abr_env_set_lane_id_base(self._h, int(lane_id_base)))
        impls = {"jump": 0, "tick": 1, "split": 2, "auto": 3, "async": 4, "split3": 5, "ring3": 6, "pair3": 7}
        if impl not in impls:
            raise ValueError("impl must be 'auto' (default: the fastest at this size, see effective_impl()), "
                             "'split' / 'split3' (role-split event-driven kernels, two / three waves per 64 lanes), "
                             "'jump' (event-driven, one thread per lane) or 'tick' ('async' / 'ring3' only with "
                             "library=<a diagnostic build that carries them>)")
        self.impl = impl
        self._check(self.lib.abr_env_set_impl(self._h, impls[impl]))
        if self.lane_speeds is not None:
            self._check(self.lib.abr_env_set_speed_schedule(self._h, _lib.ptr(self.lane_speeds),
                                                           int(self.lane_speeds.shape[0])))
        if self.br_table is not None:
            self._check(self.lib.abr_env_set_bitrate_table(self._h, _lib.ptr(self.br_table)))
        self.obs = torch.zeros(OBS_DIM, self.n_lanes, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(self.n_lanes, dtype=torch.float32, device=self.device)
        self.done = torch.zeros(self.n_lanes, dtype=torch.uint8, device=self.device)
        self.trace_id = None
        self.start_offset = None

    # -- plumbing ----------------------------------------------------------
    def _check(self, rc):
        _lib.check(rc, self.lib)

    def _stream(self):
        return _lib.current_stream(self.device)

    def _call(self, fn, *args):
        """One C-ABI call with self.device current: the library launches on the stream it is
        handed, and HIP launches go to the CURRENT device."""
        if torch.cuda.current_device() == self._dev_index:     # the common case: no context switch to pay
            self._check(fn(*args, self._stream()))
        else:
            with torch.cuda.device(self.device):
                self._check(fn(*args, self._stream()))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.abr_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _i32(self, t, name):
        t = torch.as_tensor(t, device=self.device)
        if t.dtype != torch.int32:
            t = t.to(torch.int32)
        t = t.contiguous()
        if t.shape != (self.n_lanes,):
            raise ValueError(f"{name} must have shape ({self.n_lanes},), got {tuple(t.shape)}")
        return t

    # -- the step surface --------------------------------------------------
    def reset(self, trace_id=None, start_offset=None, mask=None, check=False):
        """Simulator.py:95-133 + idle ticks to the first ABR call.  Default assignment: lane i -> trace
        i % n_traces, offset 0.  `mask`: only lanes with a non-zero byte are reset (a masked reset inside an RL loop).

        No device-to-host synchronisation happens here: a lane whose trace id is outside [0, n_traces) or whose start
        offset is negative is frozen ON THE DEVICE with ABR_DONE_BADARG in its done byte (the reference would raise
        IndexError at Simulator.py:159); the observation returned for it looks like a fresh lane's.  `done_after_reset()`
        reads the lanes' done bytes as they stand (a zero-copy view, no step needed) so that a caller can see such lanes
        without a step.  check=True validates on the host first and raises ValueError instead -- before this object or
        the device state changes -- at the price of two synchronisations."""
        if trace_id is None:
            trace_id = torch.arange(self.n_lanes, device=self.device, dtype=torch.int32) % self.n_traces
        tid = self._i32(trace_id, "trace_id")
        if start_offset is None:
            start_offset = torch.zeros(self.n_lanes, dtype=torch.int32, device=self.device)
        off = self._i32(start_offset, "start_offset")
        if check:                                  # before anything of this object changes
            if int(tid.min()) < 0 or int(tid.max()) >= self.n_traces:
                raise ValueError("trace_id out of range")
            if int(off.min()) < 0:
                raise ValueError("start_offset must be >= 0")
        self.trace_id, self.start_offset = tid, off
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self._call(self.lib.abr_env_reset, self._h, _lib.ptr(self.trace_id),
                   _lib.ptr(self.start_offset), _lib.ptr(m), _lib.ptr(self.obs))
        return self.obs

    def done_after_reset(self):
        """The lanes' done bytes as they stand in the workspace (uint8 [N], zero-copy): ABR_DONE_BADARG for a lane the last
        reset() froze because of its trace id / start offset, 0 for a lane that is running."""
        return self.mpc_inputs()[5]

    def step(self, actions):
        """One chunk per lane.  Returns (obs f32[OBS_DIM,N], reward f32[N], done u8[N]);
        the tensors are reused across calls."""
        a = self._i32(actions, "actions")
        self._call(self.lib.abr_env_step, self._h, _lib.ptr(a), _lib.ptr(self.obs),
                   _lib.ptr(self.reward), _lib.ptr(self.done))
        return self.obs, self.reward, self.done

    # the Pensieve-style name BASELINE.json uses for the same call
    get_video_chunk = step

    def step_random(self, n_steps: int, seed: int, out=None, want_actions=True):
        """n_steps fused decisions per lane under the built-in counter-based
        random policy.  Returns dict(obs[n,OBS_DIM,N], reward[n,N], done[n,N], actions[n,N])."""
        n = int(n_steps)
        if out is None:
            out = dict(
                obs=torch.empty(n, OBS_DIM, self.n_lanes, dtype=torch.float32, device=self.device),
                reward=torch.empty(n, self.n_lanes, dtype=torch.float32, device=self.device),
                done=torch.empty(n, self.n_lanes, dtype=torch.uint8, device=self.device),
                actions=(torch.empty(n, self.n_lanes, dtype=torch.int32, device=self.device)
                         if want_actions else None))
        ptrs = getattr(out, "ptrs", None)
        if ptrs is None:
            ptrs = (_lib.ptr(out.get("obs")), _lib.ptr(out.get("reward")), _lib.ptr(out.get("done")),
                    _lib.ptr(out.get("actions")))
        self._call(self.lib.abr_env_step_random, self._h, n, C.c_uint64(int(seed) & (2 ** 64 - 1)), *ptrs)
        return out

    def bind_out(self, out):
        """Look the device pointers of an output dict (obs / reward / done / actions, any of them None) up
        once; pass the result as `out=` to step_random."""
        b = BoundOut(out)
        b.ptrs = (_lib.ptr(out.get("obs")), _lib.ptr(out.get("reward")), _lib.ptr(out.get("done")),
                  _lib.ptr(out.get("actions")))
        return b

    def step_script(self, actions, out=None):
        """len(actions) fused decisions per lane with the ABR controller's answers given up front:
        actions int32 [n_steps, N] (what run() does with a scripted abr_controller,
        Simulator.py:155).  Returns dict(obs[n,OBS_DIM,N], reward[n,N], done[n,N])."""
        a = torch.as_tensor(actions, device=self.device)
        if a.dtype != torch.int32:
            a = a.to(torch.int32)
        a = a.contiguous()
        if a.dim() != 2 or a.shape[1] != self.n_lanes or a.shape[0] < 1:
            raise ValueError(f"actions must have shape (n_steps, {self.n_lanes}), got {tuple(a.shape)}")
        n = int(a.shape[0])
        if out is None:
            out = dict(
                obs=torch.empty(n, OBS_DIM, self.n_lanes, dtype=torch.float32, device=self.device),
                reward=torch.empty(n, self.n_lanes, dtype=torch.float32, device=self.device),
                done=torch.empty(n, self.n_lanes, dtype=torch.uint8, device=self.device))
        self._call(self.lib.abr_env_step_script, self._h, n, _lib.ptr(a), _lib.ptr(out.get("obs")),
                   _lib.ptr(out.get("reward")), _lib.ptr(out.get("done")))
        return out

    def effective_impl(self, fused: bool = False):
        """Name of the kernels the handle resolves to right now ('auto' is a policy, not a kernel): fused=True for a
        step_random / step_script call of MORE than one decision, False for launches of one decision (step, step_mpc,
        and a fused call with n_steps == 1) -- under 'auto' those resolve to 'jump' at every size (include/abr_env.h)."""
        v = C.c_int32()
        self._check(self.lib.abr_env_get_effective_impl(self._h, int(bool(fused)), C.byref(v)))
        return {0: "jump", 1: "tick", 2: "split", 4: "async", 5: "split3", 6: "ring3", 7: "pair3"}[v.value]

    def step_mpc(self, controller, n_steps: int, out=None, want_obs=True, want_actions=True):
        """n_steps decisions per lane taken by `controller` (a BatchedMPCController whose
        tables match this environment) on the device: next_bitrate() on each lane's own
        state, then the download, with no host work between decisions.  Returns
        dict(obs[n,OBS_DIM,N], reward[n,N], done[n,N], actions[n,N])."""
        n = int(n_steps)
        br, sz = controller._tables()
        cfg = controller.config()
        if out is None:
            out = dict(
                obs=(torch.empty(n, OBS_DIM, self.n_lanes, dtype=torch.float32, device=self.device)
                     if want_obs else None),
                reward=torch.empty(n, self.n_lanes, dtype=torch.float32, device=self.device),
                done=torch.empty(n, self.n_lanes, dtype=torch.uint8, device=self.device),
                actions=(torch.empty(n, self.n_lanes, dtype=torch.int32, device=self.device)
                         if want_actions else None))
        self._call(self.lib.abr_env_step_mpc, self._h, C.byref(cfg), _lib.ptr(br), _lib.ptr(sz), n,
                   _lib.ptr(out.get("obs")), _lib.ptr(out.get("reward")), _lib.ptr(out.get("done")),
                   _lib.ptr(out.get("actions")))
        return out

    # -- exact state -------------------------------------------------------
    def observe_f64(self):
        """dict of float64 [N] tensors: everything the reference's run() frame holds
        at the call site (rows: _lib.F64_ROWS)."""
        out = torch.empty(F64_DIM, self.n_lanes, dtype=torch.float64, device=self.device)
        self._call(self.lib.abr_env_observe_f64, self._h, _lib.ptr(out))
        return {k: out[i] for i, k in enumerate(F64_ROWS)}

    def episode_qoe(self):
        """calculate_qoe (Simulator.py:79-86) of each lane's (last) finished episode, float64 [N]."""
        out = torch.empty(self.n_lanes, dtype=torch.float64, device=self.device)
        self._call(self.lib.abr_env_episode_qoe, self._h, _lib.ptr(out))
        return out

    def state_view(self):
        v = _lib.StateView()
        self._check(self.lib.abr_env_get_state(self._h, C.byref(v)))
        return v

    def _view(self, addr, dtype, shape):
        """Zero-copy tensor over a region of the workspace."""
        off = addr - self.workspace.data_ptr()
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        return self.workspace[off:off + nbytes].view(dtype).view(*shape)

    def history(self):
        """(previous_bitrates u8[V,N], previous_bandwidths f64[V,N]) -- rows >= chunk_id are stale."""
        v = self.state_view()
        return (self._view(v.action_hist, torch.uint8, (self.video_length, self.n_lanes)),
                self._view(v.bw_hist, torch.float64, (self.video_length, self.n_lanes)))

    def mpc_inputs(self):
        """Zero-copy float64/int32 tensors the MPC kernel reads and mutates (D9):
        (chunk_id, last_bitrate, buffer_level, hist_n, hist_sum_inv, done)."""
        v = self.state_view()
        N = self.n_lanes
        return (self._view(v.chunk_id, torch.int32, (N,)), self._view(v.last_bitrate, torch.int32, (N,)),
                self._view(v.buffer_level, torch.float64, (N,)), self._view(v.hist_n, torch.float64, (N,)),
                self._view(v.hist_sum_inv, torch.float64, (N,)), self._view(v.done, torch.uint8, (N,)))

    # -- checkpoint / resume ------------------------------------------------
    TAG_BYTES = 256     # the workspace's last 256 bytes: its layout tag (include/abr_env.h, ABI 4)

    def state_dict(self):
        """All simulator state is the workspace tensor (the reference keeps it in
        run() locals and cannot checkpoint).  The dict is stamped with the ABI version and the
        workspace size it was taken under; the workspace itself ends in the library's layout tag."""
        return dict(workspace=self.workspace.clone(), trace_id=self.trace_id, start_offset=self.start_offset,
                    abi_version=_lib.ABI_VERSION, workspace_bytes=int(self.workspace.numel()))

    def load_state_dict(self, sd):
        """Refuses -- before anything is copied -- a state that was taken under another ABI version (the lane-state layout
        changes between versions; sizes can coincide), with another lane count or another configuration: the stamp of
        state_dict() and the layout tag at the end of the saved workspace are compared with this handle's own."""
        if sd.get("abi_version") != _lib.ABI_VERSION:
            raise ValueError(f"state_dict was taken under ABI version {sd.get('abi_version')}, this library is version "
                             f"{_lib.ABI_VERSION}: the workspace layout differs, the checkpoint cannot be restored")
        w = sd["workspace"]
        if w.numel() != self.workspace.numel() or sd.get("workspace_bytes") != self.workspace.numel():
            raise ValueError("workspace size mismatch: different config or lane count")
        if not torch.equal(w[-self.TAG_BYTES:].cpu(), self._tag.cpu()):
            raise ValueError("workspace layout tag mismatch: the checkpoint belongs to another lane count, configuration "
                             "or library version")
        self.workspace.copy_(w)
        self.trace_id, self.start_offset = sd["trace_id"], sd["start_offset"]
        # the handle now carries episodes in flight (a freshly built one had none): the speeds /
        # bitrate table given to __init__ are in force, later setter calls are latched again
        self._check(self.lib.abr_env_notify_restore(self._h))


def obs_dict(obs):
    """Name the rows of an observation tensor."""
    return {k: obs[..., i, :] for i, k in enumerate(OBS_ROWS)}
