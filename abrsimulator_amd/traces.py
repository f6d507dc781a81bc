"""Trace / MPD file formats of the reference.

* Network trace: one float per line, `bandwidths.append(float(line))`
  (Simulator.py:59-65).  Parsed with Python's float(), so the float64 handed to
  the device is bit-identical to what the reference would hold.
* MPD file: one line per chunk holding that chunk's bitrate ladder
  (Simulator.py:68-77).  The reference's parser is broken -- `float(line.split())`
  raises TypeError for every line (D4) -- so this is the evident intent: split on
  whitespace, one float per field, one Chunk per line, video_length = line count.
"""
from typing import List, Sequence

from .datamodel import MPD, Chunk, NetworkInfo


def load_trace_file(path: str) -> List[float]:
    out = []
    with open(path) as f:
        for line in f.readlines():
            if line.strip() == "":
                continue            # the reference would raise ValueError on a blank line
            out.append(float(line))
    if not out:
        raise ValueError(f"{path}: empty trace")
    return out


def load_network_info(interval: float, paths) -> NetworkInfo:
    """set_network_info(interval, networktrace) for one path or a list of paths
    (one trace per file; lanes pick traces by index)."""
    if isinstance(paths, (str, bytes)):
        paths = [paths]
    return NetworkInfo(float(interval), [load_trace_file(p) for p in paths])


def save_trace_file(path: str, bandwidths: Sequence[float]) -> None:
    with open(path, "w") as f:
        for b in bandwidths:
            f.write(repr(float(b)) + "\n")      # repr round-trips float64 exactly


def load_mpd_file(chunk_length: float, max_buffer: float, start_up_length: float, path: str,
                  sizes_from_bitrate: bool = True) -> MPD:
    """set_mpd(chunk_length, max_buffer, start_up_length, mpdfile), repaired."""
    chunks = []
    with open(path) as f:
        for line in f.readlines():
            fields = line.split()
            if not fields:
                continue
            br = [float(x) for x in fields]
            chunks.append(Chunk(br, [b * chunk_length for b in br] if sizes_from_bitrate else None))
    if not chunks:
        raise ValueError(f"{path}: empty MPD")
    n = len(chunks[0].bitrates)
    if any(len(c.bitrates) != n for c in chunks):
        raise ValueError(f"{path}: every chunk needs the same number of bitrates")
    return MPD(len(chunks), chunk_length, max_buffer, start_up_length, chunks)


def save_mpd_file(path: str, ladders: Sequence[Sequence[float]]) -> None:
    with open(path, "w") as f:
        for lad in ladders:
            f.write(" ".join(repr(float(b)) for b in lad) + "\n")
