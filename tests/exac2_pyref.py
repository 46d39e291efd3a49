"""Pure-Python decoder of an EXAC v2 chunk stream, written from the format text of DESIGN.md 3.11b
(not from oracle/exac_codec.c): an independent reading of the specification that the committed
vectors, the C restatement and the HIP kernels all have to satisfy.  Loops over elements -- small
inputs only."""
import numpy as np

EDGES = (1, 2, 3, 4, 5, 6, 8, 10, 13, 17, 22, 30, 45, 70, 120)
L, M = 1 << 15, 1 << 12


def decode(stream):
    b = bytes(stream)
    assert b[:3] == b"EX\x02"
    ts = b[3]
    n, ey, ex, nwords = (int.from_bytes(b[4 + 4 * k:8 + 4 * k], "little") for k in range(4))
    present = [int.from_bytes(b[20 + 8 * c:28 + 8 * c], "little") for c in range(16)]
    wide = [int.from_bytes(b[148 + 8 * c:156 + 8 * c], "little") for c in range(16)]
    pos = 276
    cum = []
    for c in range(16):
        syms = [s for s in range(64) if present[c] >> s & 1]
        wsyms = [s for s in syms if wide[c] >> s & 1]
        low = b[pos:pos + len(syms)]
        high = b[pos + len(syms):pos + len(syms) + len(wsyms)]
        pos += len(syms) + len(wsyms)
        F = [0] * 64
        for k, s in enumerate(syms):
            F[s] = low[k] + 1
        for k, s in enumerate(wsyms):
            F[s] += high[k] << 8
        assert not syms or sum(F) == M
        acc, row = 0, []
        for s in range(64):
            row.append(acc)
            acc += F[s]
        cum.append((F, row))
    pos += pos & 1
    words = [int.from_bytes(b[pos + 2 * k:pos + 2 * k + 2], "little") for k in range(nwords)]
    assert pos + 2 * nwords == len(b)
    cursor = nwords - 128 if nwords else 0
    x = [L] * 64
    if nwords:
        x = [words[cursor + 2 * l] | words[cursor + 2 * l + 1] << 16 for l in range(64)]
    plane = ey * ex
    mag = [0] * n
    val = [0] * n

    def refill(lanes):
        nonlocal cursor
        cursor -= len(lanes)
        for k, l in enumerate(lanes):
            x[l] = (x[l] << 16) | words[cursor + k]

    for r0 in range(0, n, 64):
        lanes = range(min(64, n - r0))
        info = {}
        for l in lanes:
            i = r0 + l
            y, z = (i // ex) % ey, i // plane
            ku, kb = l // ex + 1, l // plane + 1
            U = y >= ku and ku * ex <= 8000
            B = z >= kb and kb * plane <= 8000
            ju, jb = i - ku * ex, i - kb * plane
            a = mag[ju] + mag[jb] if U and B else 2 * mag[ju] if U else 2 * mag[jb] if B else 0
            ctx = sum(1 for e in EDGES if e <= a)
            pred = 0
            if ts == 2:
                pred = (val[ju] + val[jb] + 1) >> 1 if U and B else val[ju] if U else val[jb] if B else 0
            F, C = cum[ctx]
            slot = x[l] & (M - 1)
            s = max(t for t in range(64) if F[t] and C[t] <= slot)
            x[l] = F[s] * (x[l] >> 12) + slot - C[s]
            info[l] = [s, 0 if s < 32 else s - 30, 0, pred]
        refill([l for l in lanes if x[l] < L])
        for j in range(3):
            todo = [l for l in lanes if info[l][1] > 12 * j]
            for l in todo:
                k = min(info[l][1] - 12 * j, 12)
                f = M >> k
                slot = x[l] & (M - 1)
                info[l][2] |= (slot >> (12 - k)) << (12 * j)
                x[l] = f * (x[l] >> 12) + (slot & (f - 1))
            refill([l for l in todo if x[l] < L])
        for l in lanes:
            s, nb, e, pred = info[l]
            u = s if s < 32 else 32 + (((1 << (s - 32)) - 1) << 2) + e
            mag[r0 + l] = min((u + 1) >> 1, 127)
            r = (u >> 1) ^ -(u & 1)
            val[r0 + l] = (pred + r) & 0xFFFF if ts == 2 else r
    assert cursor == 0
    return np.array(val, dtype=np.uint16 if ts == 2 else np.int64).astype(np.uint16 if ts == 2 else np.int32), (n // plane, ey, ex)
