"""TEST INFRASTRUCTURE ONLY - numpy statement of the counter-based RNG used by the HIP kernels.

The reference draws dropout masks from torch's global generator (``nn.Dropout``,
baseline/models/CNN.py:59-61, CRNN.py:26,74) and the teacher noise from ``np.random.normal``
(baseline/DataLoad.py:283-285).  Those streams cannot be reproduced on a GPU, so the HIP path
defines its own stream (Philox4x32-10, Salmon et al. SC'11) and the oracle takes the resulting
masks / noise as explicit inputs.  This file states that stream so train-mode parity tests can run
with dropout ON: same seed -> bit-identical masks on both sides.

Stream definition (mirrored in dcase2019_task4_amd/csrc/philox.h):

  key      = (seed & 0xffffffff, seed >> 32)
  counter  = (index, 0, stream_id, 0x5ED0)
  out[4]   = philox4x32_10(counter, key);  byte i (0..15) = (out[i>>2] >> (8*(i&3))) & 0xff
  keep(b)  = b >= thr, thr = round(p * 256);  kept values are scaled by 256/(256 - thr)
             (the exact keep probability of an 8-bit draw; p = 0.5, 0.25 are represented exactly)

* conv-block dropout, block l (0,1,2), tensor laid out [B][H][W][C] with a (2,4) pooling window
  behind it:  q = (b*Ho + h//2)*Wo + w//4,  dt = h & 1,  df = w & 3,
              index = (q >> 2)*C + c,  stream_id = 2*l + dt,  byte = (q & 3)*4 + df.
  Rows h >= 2*Ho (odd H, dropped by the floor-mode pool) get no mask (value irrelevant; 0 here).
  p == 0.5 (thr == 128, the reference's rate) uses ONE bit per element instead: with rb = q >> 2,
              lane = dt*32 + (c & 31),  unit u = rb*(C/32) + (c >> 5),  index = (u >> 3)*64 + lane,  stream_id = 32 + l,
              field = u & 7  (16-bit field f = (word[f>>1] >> 16*(f&1)) & 0xffff),
              keep = bit (q & 3)*4 + df of the field; kept values are scaled by 2.
              (C = 64: index = (rb >> 2)*64 + lane, field = (rb & 3)*2 + (c >> 5).)
* recurrent-output dropout, tensor [B][T][2H] flattened to e: index = e >> 4, stream_id = 8,
  byte = e & 15.
* teacher noise, tensor [frames][n_mels] flattened to e per clip b: two 32-bit uniforms from
  index = (b*frames*n_mels + e) >> 1, stream_id = 16, words (2*(e&1), 2*(e&1)+1) ->
  Box-Muller normal -> |0.25 * n|.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
TAG = 0x5ED0


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. All args uint32 arrays (broadcastable). Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape).copy()
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape).copy()
    c0 = c0.copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def _bytes(index, stream_id, seed):
    k0, k1 = _key(seed)
    o = philox4x32_10(index, 0, np.uint32(stream_id), np.uint32(TAG), k0, k1)
    by = np.empty(index.shape + (16,), dtype=np.uint32)
    for i in range(16):
        by[..., i] = (o[i >> 2] >> np.uint32(8 * (i & 3))) & np.uint32(0xFF)
    return by


def thresh8(p):
    return int(float(p) * 256.0 + 0.5)


def keep_scale(p):
    return np.float32(256.0) / np.float32(256.0 - thresh8(p))


def dropout_mask_pooled(seed, block, B, H, W, C, p):
    """Mask for conv-block ``block`` in NHWC [B,H,W,C] float32: 0 or 256/(256-thr)."""
    if p <= 0.0:
        return np.ones((B, H, W, C), dtype=np.float32)
    Ho, Wo = H // 2, W // 4
    out = np.zeros((B, H, W, C), dtype=np.float32)
    if thresh8(p) == 128:
        out[:, : 2 * Ho, : 4 * Wo, :] = _mask_pooled_1bit(seed, block, B, Ho, Wo, C)
        return out
    # one Philox draw per (row block of 4 pooled pixels, channel, dt): 16 bytes = 4 pooled x 4 df
    Q = B * Ho * Wo
    nrb = (Q + 3) // 4
    rb, c = np.meshgrid(np.arange(nrb), np.arange(C), indexing="ij")
    index = (rb * C + c).astype(np.uint32)
    thr = thresh8(p)
    keepv = keep_scale(p)
    res = np.zeros((B * Ho * Wo, 2, 4, C), dtype=np.float32)               # [q, dt, df, c]
    for d in (0, 1):
        by = _bytes(index, 2 * block + d, seed)                              # [nrb, C, 16]
        keep = (by >= thr).astype(np.float32) * keepv
        keep = keep.reshape(nrb, C, 4, 4).transpose(0, 2, 3, 1).reshape(nrb * 4, 4, C)[:Q]   # [q, df, c]
        res[:, d] = keep
    res = res.reshape(B, Ho, Wo, 2, 4, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, 2 * Ho, 4 * Wo, C)
    out[:, : 2 * Ho, : 4 * Wo, :] = res
    return out


def _mask_pooled_1bit(seed, block, B, Ho, Wo, C):
    """p = 0.5 stream: one bit per element, 8 (row block, 32-channel slice) units per draw. Returns [B, 2Ho, 4Wo, C]."""
    assert C % 32 == 0
    NH = C // 32
    Q = B * Ho * Wo
    nrb = (Q + 3) // 4
    n_units = nrb * NH
    n_draw = (n_units + 7) // 8
    d, lane = np.meshgrid(np.arange(n_draw), np.arange(64), indexing="ij")
    k0, k1 = _key(seed)
    o = philox4x32_10((d * 64 + lane).astype(np.uint32), 0, np.uint32(32 + block), np.uint32(TAG), k0, k1)
    fields = np.empty((n_draw, 64, 8), dtype=np.uint32)
    for f in range(8):
        fields[..., f] = (o[f >> 1] >> np.uint32(16 * (f & 1))) & np.uint32(0xFFFF)
    # [draw, lane = (dt, n), f] -> units u = draw*8 + f = rb*NH + h
    bits = ((fields[..., None] >> np.arange(16, dtype=np.uint32)) & np.uint32(1)).astype(np.float32) * np.float32(2.0)
    bits = bits.reshape(n_draw, 2, 32, 8, 4, 4)            # draw, dt, n, f, j, df
    units = bits.transpose(0, 3, 1, 2, 4, 5).reshape(n_draw * 8, 2, 32, 4, 4)[:n_units]      # u, dt, n, j, df
    units = units.reshape(nrb, NH, 2, 32, 4, 4)            # rb, h, dt, n, j, df
    keep = units.transpose(0, 4, 2, 5, 1, 3).reshape(nrb * 4, 2, 4, C)[:Q]                    # q = rb*4 + j, dt, df, c = h*32 + n
    return keep.reshape(B, Ho, Wo, 2, 4, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, 2 * Ho, 4 * Wo, C)


def dropout_mask_flat(seed, stream_id, shape, p):
    """Mask for a contiguous tensor of ``shape`` (recurrent-output dropout, stream 8)."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return np.ones(shape, dtype=np.float32)
    ncall = (n + 15) // 16
    by = _bytes(np.arange(ncall, dtype=np.uint32), stream_id, seed).reshape(-1)[:n]
    return ((by >= thresh8(p)).astype(np.float32) * keep_scale(p)).reshape(shape)


def teacher_noise(seed, B, frames, n_mels, std=0.25):
    """|N(0, std)| noise [B, frames, n_mels] float32: Box-Muller on Philox uniforms, stream 16, one draw per element PAIR.

    pair i = elements (2 i, 2 i + 1) of the flattened tensor; (w0, w1) = the first two words of Philox4x32-10 at counter i;
    u1 = (w0 + 1) * 2^-32 in (0,1],  u2 = w1 * 2^-32 in [0,1),  r = sqrt(-2 ln u1),
    element 2 i = r cos(2 pi u2), element 2 i + 1 = r sin(2 pi u2) - both outputs of the transform, float64 arithmetic,
    rounded to float32 (csrc/feat.hip teacher_noise_pair).
    """
    n = B * frames * n_mels
    npair = (n + 1) // 2
    k0, k1 = _key(seed)
    o = philox4x32_10(np.arange(npair).astype(np.uint32), 0, np.uint32(16), np.uint32(TAG), k0, k1)
    u1 = (o[0].astype(np.float64) + 1.0) * 2.0 ** -32
    u2 = o[1].astype(np.float64) * 2.0 ** -32
    r = np.sqrt(-2.0 * np.log(u1))
    g = np.stack([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)], axis=-1).reshape(-1)[:n]
    return np.abs(std * g).astype(np.float32).reshape(B, frames, n_mels)
