"""Plain attribute bags with the reference's field names and units.

These mirror the data model of the reference (Simulator.py:4-42 and the richer
variants the MPC side reads, mpc_test.py:13-37) so that code written against
the reference's objects can hand them to the batched environment unchanged.
They hold host Python numbers; the device tensors are built from them once.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence


@dataclass
class Chunk:
    """Simulator.py:4-6; `sizes` is the MPC-side extension (mpc_test.py:13-16).
    bitrates[0] is the lowest rate.  sizes defaults to bitrate * chunk_length,
    the constant-bitrate relation run() uses (Simulator.py:156)."""
    bitrates: Sequence[float]
    sizes: Optional[Sequence[float]] = None


@dataclass
class MPD:
    """Simulator.py:11-17.  video_length [chunks], chunk_length [s], max_buffer
    (compared against buffer_level in seconds, Simulator.py:190),
    start_up_length [s].  `chunks` is either ONE Chunk (the single ladder
    Simulator.run indexes, Simulator.py:82,156) or a list of per-chunk Chunks
    (what set_mpd builds and mpc.py indexes, Simulator.py:71-76, mpc.py:126)."""
    video_length: int
    chunk_length: float
    max_buffer: float
    start_up_length: float = 0.0
    chunks: object = None

    def ladder(self) -> List[float]:
        """The single ladder Simulator.run() indexes (Simulator.py:82,156).  A per-chunk
        MPD whose lines differ has no single ladder: raises instead of silently using
        chunk 0's (use bitrate_table() for the per-chunk form)."""
        if isinstance(self.chunks, (list, tuple)):
            first = [float(b) for b in self.chunks[0].bitrates]
            for i, c in enumerate(self.chunks):
                if [float(b) for b in c.bitrates] != first:
                    raise ValueError(f"MPD chunk {i} has a different bitrate ladder than chunk 0: "
                                     "this MPD has no single ladder")
            return first
        return [float(b) for b in self.chunks.bitrates]

    def uniform(self) -> bool:
        """True when every chunk carries the same bitrate ladder."""
        try:
            self.ladder()
            return True
        except ValueError:
            return False

    def bitrate_table(self) -> List[List[float]]:
        """[video_length][n_rates] bitrates, one row per chunk."""
        return [[float(b) for b in c.bitrates] for c in self.chunk_list()]

    def chunk_list(self) -> List[Chunk]:
        if isinstance(self.chunks, (list, tuple)):
            return list(self.chunks)
        return [self.chunks] * int(self.video_length)


@dataclass
class QOEMetric:
    """Simulator.py:19-24 (latency_weight is absent on the MPC side, mpc_test.py:25-29)."""
    rebuffer_weight: float
    variance_weight: float
    startup_weight: float
    latency_weight: float = 0.0


@dataclass
class NetworkInfo:
    """Simulator.py:39-42: a square wave, bandwidths[i] holds during the i-th
    interval.  Here `bandwidths` may also be a list of traces (one per row)."""
    interval: float
    bandwidths: Sequence = field(default_factory=list)


@dataclass
class ChunkInfo:
    """Simulator.py:30-35 / mpc_test.py:31-37: what the ABR plugin is told
    about the next chunk.  Batched: every field is a tensor over lanes."""
    chunk_id: object
    previous_bitrates: object
    previous_bandwidths: object
    buffer_level: object
