"""Tracking results -> benchmark submission files (the wire formats at the end of the path).

MOT16/17 text format as datasets/tracking/mot17_sequence.py:209-242 writes it, MOTS20 text format as
datasets/tracking/mots20_sequence.py:72-92 does.  The latter needs COCO's compressed run-length mask
encoding, which the reference takes from pycocotools (`rletools.encode`); pycocotools is not available
offline, so the encoding is restated here from its published algorithm (cocoapi maskApi.c: rleEncode,
rleToString, rleFrString) -- parity with pycocotools itself is unpinned, the tests use hand-derived
vectors and round trips.
"""
import csv
import os

import numpy as np


def write_mot_results(results: dict, path: str) -> None:
    """results: Tracker.get_results() -> `<frame>,<id>,<left>,<top>,<width>,<height>,-1,-1,-1,-1`
    (1-based frame, id and pixel coordinates)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="") as f:
        writer = csv.writer(f, delimiter=',')
        for track_id, track in results.items():
            for frame, data in track.items():
                x1, y1, x2, y2 = (data['bbox'][k] for k in range(4))
                writer.writerow([frame + 1, track_id + 1, x1 + 1, y1 + 1, x2 - x1 + 1, y2 - y1 + 1,
                                 -1, -1, -1, -1])


def load_mot_results(path: str) -> dict:
    """Inverse of write_mot_results: {track_id: {frame: {'bbox': np.float32[4] xyxy}}} (0-based)."""
    results = {}
    if not os.path.isfile(path):
        return results
    with open(path, newline="") as f:
        for row in csv.reader(f, delimiter=','):
            frame, track_id = int(row[0]) - 1, int(row[1]) - 1
            x1, y1 = float(row[2]) - 1, float(row[3]) - 1
            x2, y2 = x1 + float(row[4]) - 1, y1 + float(row[5]) - 1
            results.setdefault(track_id, {})[frame] = {
                'bbox': np.array([x1, y1, x2, y2], dtype=np.float32)}
    return results


# ------------------------------------------------------------------------------- COCO compressed RLE
def rle_counts(mask: np.ndarray) -> list:
    """Run lengths of the column-major flattened binary mask, starting with the run of zeros."""
    flat = np.asarray(mask, dtype=bool).reshape(-1, order='F')
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    counts = np.diff(bounds).tolist()
    return ([0] + counts) if flat[0] else counts


def rle_to_string(counts) -> bytes:
    """cocoapi rleToString: counts -> LEB128-like 6-bit characters, deltas against counts[i-2] for i > 2."""
    out = bytearray()
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return bytes(out)


def rle_from_string(s: bytes) -> list:
    """cocoapi rleFrString."""
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_encode(mask: np.ndarray) -> dict:
    """{'size': [h, w], 'counts': bytes} as pycocotools.mask.encode returns for one mask."""
    mask = np.asarray(mask)
    return {'size': [int(mask.shape[0]), int(mask.shape[1])], 'counts': rle_to_string(rle_counts(mask))}


def rle_decode(rle: dict) -> np.ndarray:
    h, w = rle['size']
    counts = rle_from_string(rle['counts'])
    flat = np.zeros(h * w, dtype=bool)
    pos, val = 0, False
    for c in counts:
        if val:
            flat[pos:pos + c] = True
        pos += c
        val = not val
    return flat.reshape((h, w), order='F')


def write_mots_results(results: dict, path: str, class_id: int = 2) -> None:
    """`<frame> <id> <class> <height> <width> <rle>` per track and frame (class 2 = pedestrian)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="") as f:
        writer = csv.writer(f, delimiter=' ')
        for track_id, track in results.items():
            for frame, data in track.items():
                mask = np.asarray(data['mask'])
                writer.writerow([frame + 1, track_id + 1, class_id, mask.shape[0], mask.shape[1],
                                 rle_encode(mask)['counts'].decode('utf-8')])
