"""`Simulator`: the reference's class surface (Simulator.py:45-93) over the batched
GPU environment, so a script written against the reference keeps its calls:

    sim = Simulator(abr_controller, speed_controller, n_lanes=4096)
    sim.set_qoe_metric(QOEMetric(4.3, 1, 1, 0.1))
    sim.set_network_info(1.0, "trace.txt")            # or a list of files
    sim.set_mpd(4, 20, 8, "video.mpd")
    qoe = sim.run()                                    # float64 tensor, one QoE per lane

`abr_controller.get_next_bitrate(chunk_id, previous_bitrates, previous_bandwidths,
buffer_level)` keeps the reference's signature (Simulator.py:155) with tensors over
lanes: chunk_id int32[N]; previous_bitrates uint8[V, N] and previous_bandwidths
float64[V, N] (rows < chunk_id are the lists' contents); buffer_level float64[N].
It returns int32[N] bitrate indices.  `speed_controller.get_next_speed()` keeps the
reference's meaning -- the speed of the next played chunk (Simulator.py:176-177) -- and is asked
video_length times before the run (see Simulator._speeds).
"""
import torch

from . import _lib
from .datamodel import MPD, NetworkInfo, QOEMetric
from .env import BatchedABREnv
from .traces import load_mpd_file, load_network_info


class _UnitSpeed:
    def get_next_speed(self):
        return 1.0


class Simulator(object):
    def __init__(self, AbrController, SpeedController=None, n_lanes=1, device="cuda"):
        self.qoe_metric = None
        self.mpd = None
        self.network_info = None
        self.abr_controller = AbrController
        self.speed_controller = SpeedController if SpeedController is not None else _UnitSpeed()
        self.n_lanes = int(n_lanes)
        self.device = device
        self.env = None
        self.trace_id = None
        self.start_offset = None

    def set_qoe_metric(self, qoe_metric: QOEMetric):
        self.qoe_metric = qoe_metric

    def set_network_info(self, interval, networktrace):
        """networktrace: a trace file, a list of trace files, or a NetworkInfo."""
        if isinstance(networktrace, NetworkInfo):
            self.network_info = networktrace
        else:
            self.network_info = load_network_info(interval, networktrace)

    def set_mpd(self, chunk_length, max_buffer, start_up_length, mpdfile):
        """mpdfile: an MPD file (one ladder per line) or an MPD object."""
        if isinstance(mpdfile, MPD):
            self.mpd = mpdfile
        else:
            self.mpd = load_mpd_file(chunk_length, max_buffer, start_up_length, mpdfile)

    def set_lanes(self, trace_id=None, start_offset=None):
        """Which trace (and which offset into it) each lane plays; default lane i -> trace i % n."""
        self.trace_id, self.start_offset = trace_id, start_offset

    def get_mpd(self):
        return self.mpd          # the reference returns an undefined name here (Simulator.py:88-89)

    def get_qoe_metric(self):
        return self.qoe_metric

    def calculate_qoe(self):
        """calculate_qoe (Simulator.py:79-86) of the finished episodes."""
        return self.env.episode_qoe()

    def _speeds(self):
        """The reference asks speed_controller.get_next_speed() at the first playing tick of
        every played chunk (Simulator.py:176-177); a video has at most video_length of them,
        so the controller is asked that many times up front and its answers -- floats, or
        tensors over lanes -- become the device-side schedule (one row per played chunk).
        A controller that always answers the same float runs on the cheaper constant path."""
        answers = [self.speed_controller.get_next_speed() for _ in range(int(self.mpd.video_length))]
        if all(not torch.is_tensor(a) and not hasattr(a, "__len__") for a in answers):
            if all(float(a) == float(answers[0]) for a in answers):
                return float(answers[0])
        rows = [torch.as_tensor(a, dtype=torch.float64).reshape(-1).expand(self.n_lanes) for a in answers]
        return torch.stack(rows)

    def run(self):
        self.env = BatchedABREnv(self.mpd, self.qoe_metric, self.network_info, self.n_lanes,
                                 device=self.device, speed=self._speeds())
        env = self.env
        env.reset(self.trace_id, self.start_offset)
        chunk_id, _, buffer_level, _, _, done = env.mpc_inputs()
        prev_bitrates, prev_bandwidths = env.history()
        for _ in range(env.video_length):
            a = self.abr_controller.get_next_bitrate(chunk_id, prev_bitrates, prev_bandwidths,
                                                     buffer_level)
            env.step(torch.as_tensor(a, device=env.device).to(torch.int32))
        if not bool((done == _lib.DONE_EPISODE).all()):
            bad = int((done != _lib.DONE_EPISODE).sum())
            raise RuntimeError(f"{bad} lanes did not finish cleanly in video_length decisions "
                               "(bad action, bad reset argument or max_ticks timeout)")
        return env.episode_qoe()
