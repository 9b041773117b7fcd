"""Synthetic particle clouds of SURVEY.md section 8(d), generated on the device with torch (plumbing for bench.py and the
tests; nothing here is part of the product).

  uniform     x, y, z ~ U[0, 1), h = 0.6 (300 / (4 pi N))^(1/3)   (about 100 neighbours inside 2h)
  plummer     the recipe of the reference's test/coord_samples/plummer.hpp:17-79 RESTATED: the drand48 stream seeded with
              srand48(42), R = 1 / sqrt(u^(-2/3) - 1) accepted while R < 100, Z = (1 - 2u) R, theta = 2 pi u, scale
              3 pi / 16, centre of mass moved to the origin.  No clamping shell, no hand-set box.  The 48-bit linear
              congruential generator of drand48 (X' = 0x5DEECE66D X + 0xB mod 2^48, POSIX) has a closed-form jump, so the
              whole stream is produced in parallel; the serial part -- which stream positions start an attempt, given the
              rare rejections -- is a host walk over the ~1.5e-4 N rejected positions.  (The tests compare it with a
              serial C restatement that is pinned by the reference's own header.)
  clustered   the mixture of 8 Gaussian blobs (sigma = L / 40) at fixed centres of BASELINE configs[4], clamped to the box
              (the reference's clamped normal_distribution, test/coord_samples/random.hpp:159-174, eight times)

h of the non-uniform clouds: half the radius of the sphere that holds 100 particles at the cloud's LOCAL density (closed
form; SURVEY 8(d): "a density-scaled closed form at 1e8").
"""
import math

A48, C48, MASK48 = 0x5DEECE66D, 0xB, (1 << 48) - 1


def _lcg_jump(k):
    """(A, C) with X_{n+k} = A X_n + C mod 2^48"""
    a, c, A, C = A48, C48, 1, 0
    while k:
        if k & 1:
            A, C = (a * A) & MASK48, (a * C + c) & MASK48
        a, c = (a * a) & MASK48, (a * c + c) & MASK48
        k >>= 1
    return A, C


class Drand48Stream:
    """drand48() values number first .. first + count - 1 (0-based) of the stream behind srand48(seed), on `device`"""

    def __init__(self, seed, device, block=1 << 22):
        import torch

        self.torch, self.device, self.block = torch, device, block
        self.x0 = ((seed & 0xFFFFFFFF) << 16) | 0x330E
        # A_i, C_i for i = 1 .. block (state after i steps from a block's start state), by doubling
        A = torch.ones(block, dtype=torch.int64, device=device)
        Cc = torch.zeros(block, dtype=torch.int64, device=device)
        A[0], Cc[0] = A48, C48
        m = 1
        while m < block:
            Am, Cm = _lcg_jump(m)
            hi = min(2 * m, block)
            A[m:hi] = (A[: hi - m] * Am) & MASK48
            Cc[m:hi] = (Cc[: hi - m] * Am + Cm) & MASK48
            m *= 2
        self.A, self.C = A, Cc

    def values(self, first, count):
        """float64 tensor of drand48() results number first .. first + count - 1"""
        torch = self.torch
        out = torch.empty(count, dtype=torch.float64, device=self.device)
        done = 0
        while done < count:
            m = min(self.block, count - done)
            Aj, Cj = _lcg_jump(first + done)
            start = (Aj * self.x0 + Cj) & MASK48  # state BEFORE draw number first + done
            st = (self.A[:m] * start + self.C[:m]) & MASK48
            out[done:done + m] = st.to(torch.float64) * (1.0 / float(1 << 48))
            done += m
        return out


def plummer_reference(n, device, dtype=None, first=0, count=None, chunk=1 << 24, centre=True):
    """particles first .. first + count - 1 of the reference's n-particle Plummer sphere (plummer.hpp:17-79 restated).
    Returns x, y, z (float64 unless dtype says otherwise; the centre of mass of ALL n particles is subtracted)."""
    import torch

    count = n - first if count is None else count
    stream = Drand48Stream(42, device)
    conv = 3.0 * math.pi / 16.0
    xs, ys, zs = [], [], []
    com = torch.zeros(3, dtype=torch.float64, device=device)
    pos, made = 0, 0  # stream position of the next attempt, particles accepted so far
    while made < n:
        want = min(chunk, n - made)
        span = 3 * want + 64  # draws looked at in this round
        d = stream.values(pos, span)
        R = 1.0 / (d.pow(-2.0 / 3.0) - 1.0).sqrt()
        # an attempt that starts at stream offset p is rejected when R(d_p) is not < 100; it then consumes ONE draw,
        # an accepted attempt three.  Walk the (rare) rejected offsets on the host to find where attempts start.
        rej = torch.nonzero(~(R < 100.0)).flatten().cpu().tolist()
        starts_seg = []  # (first offset, number of accepted attempts) per run of stride-3 attempts
        p, got, ri = 0, 0, 0
        while got < want:
            kmax = min(want - got, (span - p) // 3)  # attempts whose three draws lie inside the span
            if kmax == 0:
                break  # the span is used up: the next round continues at p
            while ri < len(rej) and rej[ri] < p:
                ri += 1
            t = ri  # next rejected offset q >= p in phase with p (an entry out of phase may matter after the next shift)
            while t < len(rej) and (rej[t] - p) % 3 != 0:
                t += 1
            q = rej[t] if t < len(rej) else None
            if q is None or (q - p) // 3 >= kmax:
                starts_seg.append((p, kmax))
                p += 3 * kmax
                got += kmax
            else:
                k = (q - p) // 3
                if k:
                    starts_seg.append((p, k))
                got += k
                p = q + 1
        idx = torch.cat([torch.arange(k, device=device, dtype=torch.int64) * 3 + s for s, k in starts_seg]) \
            if starts_seg else torch.empty(0, dtype=torch.int64, device=device)
        Rk = R[idx]
        Z = (1.0 - 2.0 * d[idx + 1]) * Rk
        theta = 2.0 * math.pi * d[idx + 2]
        rad = (Rk * Rk - Z * Z).sqrt()
        X, Y = rad * theta.cos() * conv, rad * theta.sin() * conv
        Z = Z * conv
        com += torch.stack([X.sum(), Y.sum(), Z.sum()])
        lo, hi = max(first, made), min(first + count, made + got)
        if hi > lo:
            xs.append(X[lo - made:hi - made].clone())
            ys.append(Y[lo - made:hi - made].clone())
            zs.append(Z[lo - made:hi - made].clone())
        made += got
        pos += p
        del d, R, idx, Rk, Z, theta, rad, X, Y
    com = com / float(n) if centre else com * 0.0
    x = torch.cat(xs) - com[0]
    y = torch.cat(ys) - com[1]
    z = torch.cat(zs) - com[2]
    if dtype is not None:
        x, y, z = x.to(dtype), y.to(dtype), z.to(dtype)
    return x, y, z


def plummer_h(x, y, z, n_global, ng=100.0):
    """half the radius of the sphere that holds ng particles at the local density of the Plummer model (scale radius
    a = 3 pi / 16, total mass = n_global particles)"""
    a = 3.0 * math.pi / 16.0
    r2 = (x.double() ** 2 + y.double() ** 2 + z.double() ** 2) / (a * a)
    rho = 3.0 * n_global / (4.0 * math.pi * a ** 3) * (1.0 + r2).pow(-2.5)
    return (0.5 * (3.0 * ng / (4.0 * math.pi * rho)).pow(1.0 / 3.0)).to(x.dtype)


BLOB_CENTRES = [(0.25, 0.25, 0.25), (0.75, 0.25, 0.30), (0.30, 0.75, 0.25), (0.70, 0.72, 0.28),
                (0.28, 0.30, 0.75), (0.72, 0.28, 0.70), (0.25, 0.70, 0.72), (0.75, 0.75, 0.75)]


def clustered(n, device, dtype, seed, n_global=None, sigma=1.0 / 40.0):
    """mixture of 8 Gaussian blobs (sigma = L / 40) at fixed centres inside [0, 1]^3, clamped to the box"""
    import torch

    n_global = n if n_global is None else n_global
    g = torch.Generator(device=device).manual_seed(seed)
    which = torch.randint(0, 8, (n,), device=device, generator=g)
    centres = torch.tensor(BLOB_CENTRES, dtype=torch.float64, device=device)
    cols = []
    for d in range(3):
        v = centres[which, d] + sigma * torch.randn(n, dtype=torch.float64, device=device, generator=g)
        cols.append(v.clamp_(0.0, 1.0).to(dtype))
        del v
    x, y, z = cols
    # local density of the mixture -> h (the clamped tails pile up on the box faces; h there is the formula's)
    rho = torch.zeros(n, dtype=torch.float64, device=device)
    norm = n_global / 8.0 / ((2.0 * math.pi) ** 1.5 * sigma ** 3)
    for c in BLOB_CENTRES:
        d2 = (x.double() - c[0]) ** 2 + (y.double() - c[1]) ** 2 + (z.double() - c[2]) ** 2
        rho += norm * torch.exp(-0.5 * d2 / (sigma * sigma))
        del d2
    h = (0.5 * (3.0 * 100.0 / (4.0 * math.pi * rho.clamp_min_(1e-300))).pow(1.0 / 3.0)).clamp_(max=0.05).to(dtype)
    return x, y, z, h


def uniform(n, device, dtype, seed, n_global=None):
    import torch

    n_global = n if n_global is None else n_global
    g = torch.Generator(device=device).manual_seed(seed)
    x, y, z = [torch.rand(n, dtype=dtype, device=device, generator=g) for _ in range(3)]
    h0 = 0.6 * (3.0 * 100 / (4 * math.pi * n_global)) ** (1.0 / 3.0)
    return x, y, z, torch.full((n,), h0, dtype=dtype, device=device)


def make_cloud(dist, n_local, n_global, device, dtype, seed, rank=0, world=1):
    """x, y, z, h of this rank's share (a random 1 / world of the global cloud) and the box the domain is created with
    (open boundaries: the domain measures the tight box itself; the limits here are only its starting value)"""
    if dist == "uniform":
        x, y, z, h = uniform(n_local, device, dtype, seed + rank, n_global)
        return x, y, z, h, [0.0, 1.0] * 3
    if dist == "clustered":
        x, y, z, h = clustered(n_local, device, dtype, seed + rank, n_global)
        return x, y, z, h, [0.0, 1.0] * 3
    if dist == "plummer":
        # the reference's sequence is in random order already: a contiguous slice of it is a random 1 / world
        x, y, z = plummer_reference(n_global, device, dtype, first=rank * n_local, count=n_local)
        h = plummer_h(x, y, z, n_global)
        return x, y, z, h, [-1.0, 1.0] * 3
    raise ValueError(f"unknown cloud '{dist}'")
