"""Independent numpy restatement of the BM4D SPECIFICATION, written from the
text of 3.1-3.8 and not from oracle/exabm4d_oracle.c -- a second reading that the C oracle (and through
it the HIP kernels) has to agree with.  Small volumes only (vectorised per reference block).

What is bit-exact by construction: the reference grid, the candidate set, the block distance (the
fp32 fmaf chains of 3.3 are emulated exactly: a product of two fp32 numbers is exact in float64 and
`float32(float64(t) * t + acc)` rounds once, up to double-rounding cases of probability ~2^-29), the
packed keys, the admission bound and the selection (3.4).  What is held to a tolerance: the
collaborative filtering, which is evaluated here from the DEFINITIONS -- float64 DCT-II (x) Haar
matrices, float64 aggregation -- while 3.5 fixes particular fp32 operation orders; a coefficient within
1e-6 of the threshold may be kept by one and dropped by the other."""
import numpy as np

BLK, STEP, RAD, MAXG = 8, 4, 5, 16


def grid(n):
    """3.1: 0, 4, 8, ... while p + 8 <= n, plus n - 8 when (n - 8) mod 4 != 0."""
    if n < BLK:
        return []
    p = list(range(0, n - BLK + 1, STEP))
    if (n - BLK) % STEP:
        p.append(n - BLK)
    return p


def code(dz, dy, dx):
    return 0 if (dz, dy, dx) == (0, 0, 0) else 1 + ((dz + RAD) * 11 + (dy + RAD)) * 11 + (dx + RAD)


def f32_fma_chain(a, b):
    """acc = +0; for every element in raster order: t = a - b (fp32); acc = fma(t, t, acc).
    a, b: [..., 64] float32 (a cell's 4^3 voxels in z, y, x raster order); vectorised over the rest."""
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for i in range(a.shape[-1]):
        t = (a[..., i] - b[..., i]).astype(np.float32)
        acc = (t.astype(np.float64) * t.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def cells(block):
    """[..., 8, 8, 8] -> [..., 2, 2, 2, 64]: the 2 x 2 x 2 cells of 4^3 voxels, each in raster order."""
    s = block.shape[:-3]
    b = block.reshape(s + (2, 4, 2, 4, 2, 4))
    b = np.moveaxis(b, [-6, -4, -2, -5, -3, -1], [-6, -5, -4, -3, -2, -1])      # kz ky kx | z y x
    return b.reshape(s + (2, 2, 2, 64))


def distance(ref, cand):
    """3.3: S = ((C000+C001)+(C010+C011)) + ((C100+C101)+(C110+C111)), plain fp32 adds."""
    c = f32_fma_chain(cells(ref), cells(cand))                                    # [..., 2, 2, 2]
    f = np.float32
    lo = (c[..., 0, 0, 0] + c[..., 0, 0, 1]).astype(f) + (c[..., 0, 1, 0] + c[..., 0, 1, 1]).astype(f)
    hi = (c[..., 1, 0, 0] + c[..., 1, 0, 1]).astype(f) + (c[..., 1, 1, 0] + c[..., 1, 1, 1]).astype(f)
    return (lo.astype(f) + hi.astype(f)).astype(f)


def keymax(sigma, c_match):
    """3.4: (bits(fl(c_match * sigma^2 * 512)) & 0xFFFFF800) + 0x800, product in fp64, rounded once."""
    tau = np.float32(np.float64(np.float32(c_match)) * np.float64(np.float32(sigma)) * np.float64(np.float32(sigma)) * 512.0)
    return (int(tau.view(np.uint32)) & 0xFFFFF800) + 0x800


def blockmatch(vol, sigma, c_match):
    """-> keys [gz, gy, gx, 16] uint32 (unused slots 0xFFFFFFFF)."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    nz, ny, nx = vol.shape
    gz, gy, gx = grid(nz), grid(ny), grid(nx)
    kmax = keymax(sigma, c_match)
    out = np.full((len(gz), len(gy), len(gx), MAXG), 0xFFFFFFFF, dtype=np.uint32)
    for iz, rz in enumerate(gz):
        for iy, ry in enumerate(gy):
            for ix, rx in enumerate(gx):
                disp = [(dz, dy, dx) for dz in range(-RAD, RAD + 1) for dy in range(-RAD, RAD + 1)
                        for dx in range(-RAD, RAD + 1)
                        if 0 <= rz + dz <= nz - BLK and 0 <= ry + dy <= ny - BLK and 0 <= rx + dx <= nx - BLK]
                ref = vol[rz:rz + 8, ry:ry + 8, rx:rx + 8]
                cand = np.stack([vol[rz + dz:rz + dz + 8, ry + dy:ry + dy + 8, rx + dx:rx + dx + 8]
                                 for dz, dy, dx in disp])
                S = distance(np.broadcast_to(ref, cand.shape), cand)
                keys = (S.view(np.uint32) & np.uint32(0xFFFFF800)) | np.array([code(*d) for d in disp], np.uint32)
                keys = np.sort(keys[keys < kmax])[:MAXG]
                out[iz, iy, ix, :len(keys)] = keys
    return out


def decode(keys16):
    """valid displacement list of one reference block, in table order"""
    out = []
    for k in keys16:
        if k == 0xFFFFFFFF:
            break
        c = int(k) & 0x7FF
        if c == 0:
            out.append((0, 0, 0))
        else:
            c -= 1
            out.append((c // 121 - RAD, (c // 11) % 11 - RAD, c % 11 - RAD))
    return out


def dct_matrix():
    n = np.arange(8)
    D = np.sqrt(2.0 / 8.0) * np.cos(np.pi * (2 * n[None, :] + 1) * n[:, None] / 16.0)
    D[0] /= np.sqrt(2.0)
    return D


def haar_matrix(K):
    if K == 1:
        return np.ones((1, 1))
    c = 1.0 / np.sqrt(2.0)
    a = np.zeros((K // 2, K))
    d = np.zeros((K // 2, K))
    for i in range(K // 2):
        a[i, 2 * i] = a[i, 2 * i + 1] = c
        d[i, 2 * i], d[i, 2 * i + 1] = c, -c
    return np.vstack([haar_matrix(K // 2) @ a, d])


def window(beta=2.0):
    k = np.kaiser(8, beta)
    return k[:, None, None] * k[None, :, None] * k[None, None, :]


def stage(noisy, keys, sigma, basic=None, lam=2.7, beta=2.0):
    """3.5-3.8 from the definitions, float64: -> (num, den)."""
    noisy = np.asarray(noisy, dtype=np.float64)
    nz, ny, nx = noisy.shape
    gz, gy, gx = grid(nz), grid(ny), grid(nx)
    D, win = dct_matrix(), window(beta)
    num, den = np.zeros(noisy.shape), np.zeros(noisy.shape)
    s2 = float(np.float32(sigma)) ** 2
    thr = float(np.float32(np.float64(np.float32(lam)) * np.float64(np.float32(sigma))))
    for iz, rz in enumerate(gz):
        for iy, ry in enumerate(gy):
            for ix, rx in enumerate(gx):
                disp = decode(keys[iz, iy, ix])
                K = 1
                while K * 2 <= len(disp):
                    K *= 2
                disp = disp[:K]
                H = haar_matrix(K)

                def spectrum(v):
                    g = np.stack([v[rz + dz:rz + dz + 8, ry + dy:ry + dy + 8, rx + dx:rx + dx + 8] for dz, dy, dx in disp])
                    return np.einsum("kj,ua,vb,wc,jabc->kuvw", H, D, D, D, g)

                Z = spectrum(noisy)
                if basic is None:
                    keep = np.abs(Z) >= thr
                    Z = Z * keep
                    w = 1.0 / max(int(keep.sum()), 1)
                else:
                    Y = spectrum(np.asarray(basic, dtype=np.float64))
                    W = Y * Y / (Y * Y + s2)
                    Z = W * Z
                    w = 1.0 / max(float((W * W).sum()), 1.0)
                est = np.einsum("kj,ua,vb,wc,kuvw->jabc", H, D, D, D, Z)
                for j, (dz, dy, dx) in enumerate(disp):
                    sl = (slice(rz + dz, rz + dz + 8), slice(ry + dy, ry + dy + 8), slice(rx + dx, rx + dx + 8))
                    num[sl] += w * win * est[j]
                    den[sl] += w * win
    return num, den


def bm4d(vol, sigma, stages=2, c_ht=3.0, c_wie=0.6):
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    num, den = stage(vol, blockmatch(vol, sigma, c_ht), sigma)
    basic = (num / den).astype(np.float32)
    if stages == 1:
        return basic
    num, den = stage(vol, blockmatch(basic, sigma, c_wie), sigma, basic=basic)
    return (num / den).astype(np.float32)
