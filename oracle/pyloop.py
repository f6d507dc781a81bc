"""Pure-Python restatement of the tick loop -- TEST INFRASTRUCTURE.

Exists for one purpose: a like-for-like *interpreter* baseline.  BASELINE.json
asks for "the reference's own Python/CPU loop timed on the GPU box's host
cores"; the reference's files cannot travel to the GPU box, so bench.py times
this restatement (CPython float arithmetic, one tick per loop trip, exactly the
operation sequence of Simulator.py:135-208 under R1-R3) next to the C oracle.
It is written as a reset()/step() state machine -- the build's step surface --
rather than the reference's monolithic run(); tests/test_oracle_golden.py pins
it bit-for-bit to the same reference-generated fixtures as the C oracle.

Parity status: pinned.
"""


class PyTickEnv:
    """One lane.  step(action) returns (obs_dict, done)."""

    DT = 0.01  # Simulator.py:133

    def __init__(self, ladder, chunk_length, video_length, max_buffer, start_up_length, interval,
                 weights, trace, offset=0, speed=1.0):
        self.ladder = list(ladder)
        self.L = chunk_length
        self.V = video_length
        self.max_buffer = max_buffer
        self.start_up_length = start_up_length
        self.interval = interval
        self.wr, self.wv, self.ws, self.wl = weights
        self.trace = list(trace)
        self.offset = offset
        self.speed = speed

    def reset(self):
        # Simulator.py:95-130
        self.chunk = 0
        self.hist_rates = []
        self.hist_bw = []
        self.got = 0.0            # downloaded_size
        self.want = None          # target_size
        self.dl_time = 0          # download_time
        self.buf = 0              # buffer_level
        self.empty = True
        self.full = False
        self.play_id = 0
        self.play_len = 0
        self.play_time = 0
        self.inst_lat = 0
        self.avg_lat = 0
        self.starting = True
        self.t = 0.0              # global_time
        self.rebuf = 0.0
        self.startup = 0.0
        self.ticks = 0
        self.over = False
        self._head()
        while not self._at_call():
            self._tail(None)
            self._head()
        return self._obs()

    # T1-T3 of a tick (Simulator.py:137-149 with R2, R3)
    def _head(self):
        dt = self.DT
        if self.starting:
            self.startup += dt
        elif self.empty:
            self.rebuf += dt
        avail = int(self.t / self.L) - 1
        self.paused = avail < self.chunk or self.full
        self.halted = self.empty or self.starting

    def _at_call(self):
        return (not self.paused) and self.dl_time == 0

    # T4-T9 of a tick (Simulator.py:152-208)
    def _tail(self, action):
        dt = self.DT
        if not self.paused:
            if self.dl_time == 0:
                self.cur = action
                self.want = self.ladder[action] * self.L
            idx = int(self.t / self.interval)
            bw = self.trace[(self.offset + idx) % len(self.trace)]
            self.got = self.got + bw * dt
            self.dl_time += dt
            if self.got >= self.want:
                self.hist_bw.append(self.got / self.dl_time)
                self.hist_rates.append(self.cur)
                self.chunk += 1
                self.got = 0
                self.dl_time = 0
                self.buf += self.L
        if not self.halted:
            sp = self.speed
            self.inst_lat = self.t - self.play_time
            self.avg_lat = (self.avg_lat * self.play_time + self.inst_lat) / (self.play_time + sp * dt)
            self.play_time += sp * dt
            self.play_len += sp * dt
            self.buf -= sp * dt
            if self.play_len >= self.L:
                self.play_len = 0
                self.play_id += 1
        self.full = self.buf >= self.max_buffer
        if self.buf <= 0:
            self.buf = 0
            self.empty = True
        else:
            self.empty = False
        if self.starting and self.buf >= self.start_up_length:
            self.starting = False
        self.t += dt
        self.ticks += 1
        if self.chunk >= self.V:
            self.over = True

    def _obs(self):
        return dict(chunk_id=self.chunk, buffer_level=self.buf, global_time=self.t,
                    rebuffer_time=self.rebuf, start_up_time=self.startup, play_time=self.play_time,
                    average_latency=self.avg_lat, play_id=self.play_id,
                    last_bitrate=self.hist_rates[-1] if self.hist_rates else -1,
                    last_bandwidth=self.hist_bw[-1] if self.hist_bw else 0.0)

    def step(self, action):
        self._tail(action)
        while not self.over:
            self._head()
            if self._at_call():
                break
            self._tail(None)
        return self._obs(), self.over

    def qoe(self):
        # calculate_qoe, Simulator.py:79-86
        var = 0
        for i in range(self.V - 1):
            var += abs(self.ladder[self.hist_rates[i]] - self.ladder[self.hist_rates[i + 1]])
        return self.wr * self.rebuf + self.wv * var + self.ws * self.startup + self.wl * self.avg_lat


def run_episode(env, actions):
    """Replay a list of actions; returns (list of obs dicts at each call site, qoe)."""
    obs = [env.reset()]
    for a in actions:
        o, done = env.step(a)
        if not done:
            obs.append(o)
    return obs, env.qoe()
