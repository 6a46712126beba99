"""Deterministic synthetic *recorded answer streams* (the bytes a RPLIDAR sends while scanning).

Product-side input generator for the decode stage (tests, bench).  It is an ENCODER: it builds
frames of the six measurement answer types of the reference SDK
(src/sdk/include/sl_lidar_cmd.h:189-286) with correct sync nibbles / check bits and checksums,
from either random payload bits (every bit pattern a decoder can meet) or a smooth synthetic
ring, with start angles that advance like a spinning sensor, and optional corruption (bad
checksum, bad sync nibble, inserted garbage bytes, revolution-start capsules, angle jumps).
The CPU oracle is not used to make inputs.
"""
from __future__ import annotations

import zlib

import numpy as np

ANS_MEASUREMENT = 0x81
ANS_CAPSULED = 0x82
ANS_HQ = 0x83
ANS_CAPSULED_ULTRA = 0x84
ANS_DENSE_CAPSULED = 0x85
ANS_ULTRA_DENSE_CAPSULED = 0x86

FRAME_SIZE = {ANS_MEASUREMENT: 5, ANS_CAPSULED: 84, ANS_HQ: 781, ANS_CAPSULED_ULTRA: 132,
              ANS_DENSE_CAPSULED: 84, ANS_ULTRA_DENSE_CAPSULED: 170}
NODES_PER_FRAME = {ANS_MEASUREMENT: 1, ANS_CAPSULED: 32, ANS_HQ: 96, ANS_CAPSULED_ULTRA: 96,
                   ANS_DENSE_CAPSULED: 40, ANS_ULTRA_DENSE_CAPSULED: 64}
CAPSULE_TYPES = (ANS_CAPSULED, ANS_CAPSULED_ULTRA, ANS_DENSE_CAPSULED, ANS_ULTRA_DENSE_CAPSULED)
_START_ANGLE_OFF = {ANS_CAPSULED: 2, ANS_CAPSULED_ULTRA: 2, ANS_DENSE_CAPSULED: 2,
                    ANS_ULTRA_DENSE_CAPSULED: 8}


def _seal_capsules(frames: np.ndarray) -> None:
    """Write sync nibbles + XOR checksum (bytes 2..S-1) into bytes 0/1 of every frame."""
    x = np.bitwise_xor.reduce(frames[:, 2:], axis=1).astype(np.uint8)
    frames[:, 0] = 0xA0 | (x & 0x0F)
    frames[:, 1] = 0x50 | (x >> 4)


def crc32_padded(data: bytes) -> int:
    """The SDK's CRC (sl_crc.cpp): zlib CRC-32 over the data zero-padded by 4 - (len & 3)."""
    return zlib.crc32(data + b"\0" * (4 - (len(data) & 3))) & 0xFFFFFFFF


def make_frames(ans: int, n_frames: int, seed: int, *, payload: str = "random",
                frames_per_rev: float = 12.3, first_sync: bool = True,
                start_deg: float | None = None) -> np.ndarray:
    """``n_frames`` valid frames of answer type ``ans``, shape ``(n_frames, FRAME_SIZE[ans])``."""
    rng = np.random.default_rng(np.random.SeedSequence([int(seed), int(ans), 7]))
    S = FRAME_SIZE[ans]
    f = np.zeros((n_frames, S), np.uint8)
    if n_frames == 0:
        return f
    if payload == "random" or ans not in CAPSULE_TYPES:
        f[:] = rng.integers(0, 256, (n_frames, S), dtype=np.uint8)
    k = np.arange(n_frames)
    if ans in CAPSULE_TYPES:
        a0 = rng.uniform(0, 360) if start_deg is None else start_deg
        deg = (a0 + k * (360.0 / frames_per_rev)) % 360.0
        sa = (deg * 64.0).astype(np.uint16) & 0x7FFF
        if first_sync:
            sa[0] |= 0x8000
        o = _START_ANGLE_OFF[ans]
        f[:, o] = sa & 0xFF
        f[:, o + 1] = sa >> 8
        if payload in ("ring", "ring_near", "ring_noisy"):
            npf = NODES_PER_FRAME[ans]
            theta = (deg[:, None] + np.arange(npf)[None, :] * (360.0 / frames_per_rev / npf))
            r_mm = 4000.0 + 1500.0 * np.sin(np.deg2rad(theta) * 3.0 + 0.7)
            if payload in ("ring_near", "ring_noisy"):
                # a target inside the ultra-dense format's scale 0 (< 2.046 m, the only scale its
                # distance smoothing applies to): "ring" clamps that type's 2.5-5.5 m ring to a
                # CONSTANT 2046 mm.  "ring_near" is noiseless: long runs of equal 2 mm codes with an
                # occasional step, the input that keeps smoothing chains one apart longest;
                # "ring_noisy" adds the +-4 mm of range noise a real return has.
                r_mm = 1400.0 + 500.0 * np.sin(np.deg2rad(theta) * 3.0 + 0.7)
                if payload == "ring_noisy":
                    r_mm = r_mm + rng.normal(0.0, 4.0, r_mm.shape)
            drop = rng.random(r_mm.shape) < 0.08
            r_mm[drop] = 0.0
            d = r_mm.astype(np.uint32)
            if ans == ANS_DENSE_CAPSULED:
                f[:, 4::2] = (d & 0xFF).astype(np.uint8)
                f[:, 5::2] = (d >> 8).astype(np.uint8)
            elif ans == ANS_CAPSULED:
                dq2 = (d << 2).astype(np.uint32) & 0xFFFC
                d1, d2 = dq2[:, 0::2], dq2[:, 1::2]
                f[:, 4::5] = d1 & 0xFF
                f[:, 5::5] = d1 >> 8
                f[:, 6::5] = d2 & 0xFF
                f[:, 7::5] = d2 >> 8
                f[:, 8::5] = rng.integers(0, 256, (n_frames, 16), dtype=np.uint8)
            elif ans == ANS_CAPSULED_ULTRA:
                major = np.minimum(d[:, 0::3], 511)  # scale level 0 keeps the ring exact
                p1 = np.clip(d[:, 1::3].astype(np.int64) - major, -500, 500) & 0x3FF
                p2 = np.clip(d[:, 2::3].astype(np.int64) - major, -500, 500) & 0x3FF
                cx = (major.astype(np.uint32) | (p1.astype(np.uint32) << 12)
                      | (p2.astype(np.uint32) << 22))
                for b in range(4):
                    f[:, 4 + b::4] = (cx >> (8 * b)) & 0xFF
            else:  # ultra dense: scale 0 (1 mm steps of 2 mm), quality in the upper bits
                q = rng.integers(0, 256, d.shape, dtype=np.uint32)
                w = (((np.minimum(d, 2046) // 2) << 2) & 0xFFC) | ((q & 0xFF) << 12)
                e, o_ = w[:, 0::2], w[:, 1::2]
                f[:, 10::5] = e & 0xFF
                f[:, 11::5] = (e >> 8) & 0xFF
                f[:, 12::5] = o_ & 0xFF
                f[:, 13::5] = (o_ >> 8) & 0xFF
                f[:, 14::5] = ((e >> 16) & 0xF) | (((o_ >> 16) & 0xF) << 4)
        _seal_capsules(f)
    elif ans == ANS_MEASUREMENT:
        sync = (rng.random(n_frames) < 0.01).astype(np.uint8)
        f[:, 0] = (f[:, 0] & 0xFC) | sync | ((1 - sync) << 1)
        f[:, 1] |= 1
    elif ans == ANS_HQ:
        f[:, 0] = 0xA5
        for i in range(n_frames):
            c = crc32_padded(f[i, : S - 4].tobytes())
            f[i, S - 4:] = np.frombuffer(np.uint32(c).tobytes(), np.uint8)
    return f


def corrupt_stream(ans: int, frames: np.ndarray, seed: int, *, p_checksum: float = 0.03,
                   p_sync: float = 0.02, p_garbage: float = 0.02, p_revstart: float = 0.02,
                   p_jump: float = 0.02) -> np.ndarray:
    """Turn valid frames into a byte stream with the faults a serial link produces: flipped
    payload bits (checksum / CRC error), a broken sync nibble, garbage bytes between frames, and
    for the capsule types revolution-start capsules and start-angle jumps.  Returns the bytes."""
    rng = np.random.default_rng(np.random.SeedSequence([int(seed), int(ans), 11]))
    f = frames.copy()
    n, S = f.shape
    if n == 0:
        return np.zeros(0, np.uint8)
    if ans in CAPSULE_TYPES:
        o = _START_ANGLE_OFF[ans]
        rev = rng.random(n) < p_revstart
        f[rev, o + 1] |= 0x80
        jump = rng.random(n) < p_jump
        f[jump, o + 1] ^= rng.integers(1, 0x80, int(jump.sum()), dtype=np.uint8)
        _seal_capsules(f)
    bad = rng.random(n) < p_checksum
    col = rng.integers(2 if ans != ANS_HQ else 1, S, n)
    f[bad, col[bad]] ^= (1 << rng.integers(0, 8, int(bad.sum()))).astype(np.uint8)
    bsync = rng.random(n) < p_sync
    which = rng.integers(0, 2, n)
    f[bsync, which[bsync]] ^= 0xF0 if ans != ANS_MEASUREMENT else 0x01
    pieces = []
    garb = rng.random(n) < p_garbage
    for i in range(n):
        if garb[i]:
            pieces.append(rng.integers(0, 256, int(rng.integers(1, 2 * S)), dtype=np.uint8))
        pieces.append(f[i])
    return np.concatenate(pieces)


def make_stream(ans: int, n_frames: int, seed: int, *, corrupt: bool = False, **kw) -> np.ndarray:
    """One recorded stream as a flat ``uint8`` array."""
    f = make_frames(ans, n_frames, seed, **kw)
    if corrupt:
        return corrupt_stream(ans, f, seed)
    return f.reshape(-1)
