"""CPU restatement (numpy) of the device channel kernel ``ldpc_channel_llr`` — TEST INFRASTRUCTURE
ONLY.  The reference draws its noise from numpy's global generator (/root/reference/ldpc.py:251),
which no other program can reproduce, so nothing in the reference pins these values: PARITY
UNPINNED.  What this file pins is the build's own definition: Philox4x32-10 (Salmon et al., SC'11)
keyed by the seed, counter = (frame index lo, frame index hi, position // 4, 0), two Box-Muller
pairs per counter value in float32, y = x + sigma * z, llr = gain * y (ldpc.py:143-151,241-266).
The integer stream is reproduced exactly; the float32 transcendental steps to rounding."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays of uint32 counters -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for r in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        ka, kb = np.uint64((k0 + r * W0) & 0xFFFFFFFF), np.uint64((k1 + r * W1) & 0xFFFFFFFF)
        c0, c1, c2, c3 = hi1 ^ c1 ^ ka, lo1, hi0 ^ c3 ^ kb, lo0
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _box_muller(a, b):
    f32 = np.float32
    u1 = (a.astype(f32) + f32(0.5)) * f32(2.3283064365386963e-10)
    u2 = (b.astype(f32) + f32(0.5)) * f32(2.3283064365386963e-10)
    u1 = np.minimum(u1, f32(0.99999994))
    r = np.sqrt(f32(-2.0) * np.log(u1))
    ang = f32(6.283185307179586) * u2
    return r * np.cos(ang), r * np.sin(ang)


def standard_normals(frames, n, seed, first_frame=0):
    """[frames, n] float32 noise of the device kernel."""
    quads = (n + 3) // 4
    af = (np.arange(frames, dtype=np.uint64) + np.uint64(first_frame))[:, None]
    qd = np.arange(quads, dtype=np.uint64)[None, :]
    c0 = np.broadcast_to(af & MASK, (frames, quads))
    c1 = np.broadcast_to(af >> np.uint64(32), (frames, quads))
    c2 = np.broadcast_to(qd, (frames, quads))
    x, y, z, w = philox4x32_10(c0, c1, c2, np.zeros((frames, quads), np.uint64), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    n0, n1 = _box_muller(x, y)
    n2, n3 = _box_muller(z, w)
    return np.stack([n0, n1, n2, n3], axis=-1).reshape(frames, quads * 4)[:, :n]


def channel_llr(ref, frames, n, sigma2, gain, plus_is_bit, prefix_len, prefix_llr, seed, first_frame=0,
                cfo=0.0, phase0=0.0):
    """``cfo`` / ``phase0`` (cycles per symbol, cycles): the Doppler-shifted channel of
    ``ldpc_channel_llr_doppler`` — symbol i is scaled by sinc(cfo) * cos(2 pi (cfo (i + 1/2) + phase0));
    build-defined from the scale factor of /root/reference/plotDoppler.py:14-16, nothing pins it."""
    f32 = np.float32
    z = standard_normals(frames, n, seed, first_frame)
    b = np.zeros((frames, n), f32) if ref is None else np.broadcast_to(np.asarray(ref), (frames, n)).astype(f32)
    x = f32(2) * b - f32(1) if plus_is_bit else f32(1) - f32(2) * b
    if cfo or phase0:
        amp = f32(np.sinc(cfo))                                  # sin(pi cfo) / (pi cfo)
        turns = f32(cfo) * (np.arange(n, dtype=f32) + f32(0.5)) + f32(phase0)
        turns = turns - np.floor(turns)
        x = x * (amp * np.cos(f32(6.283185307179586) * turns))[None, :]
    llr = f32(gain) * (x + f32(np.sqrt(sigma2)) * z)
    llr[:, :prefix_len] = f32(prefix_llr)
    return llr


def random_message_bits(frames, k, seed, first_frame=0):
    """[frames, k] uint8 message bits of ``ldpc_encode_random``: bit i of frame a is bit (i mod 128) of
    Philox4x32-10(counter = (a lo, a hi, i // 128, 1), key = seed); word order x, y, z, w, least
    significant bit first."""
    words = (k + 127) // 128
    af = (np.arange(frames, dtype=np.uint64) + np.uint64(first_frame))[:, None]
    w = np.arange(words, dtype=np.uint64)[None, :]
    c0 = np.broadcast_to(af & MASK, (frames, words))
    c1 = np.broadcast_to(af >> np.uint64(32), (frames, words))
    c2 = np.broadcast_to(w, (frames, words))
    out = philox4x32_10(c0, c1, c2, np.ones((frames, words), np.uint64), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack(out, axis=-1).astype(np.uint32)                      # [frames, words, 4]
    bits = (u[..., None] >> np.arange(32, dtype=np.uint32)) & 1       # [frames, words, 4, 32]
    return bits.reshape(frames, words * 128)[:, :k].astype(np.uint8)
