"""The small device operations the locally essential tree gained in round 4 (csrc/let_ops.hip, extras.hip, focus.hip),
each against a numpy restatement of what the reference's host code computes at that point (GPU).  They are also run end
to end by the LET harness (oracle/let_check.cpp linked against libcstone_hip.so, tests/test_let.py -m gpu); here every
entry is called on its own through the C ABI."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    return torch


def dev(a):
    torch = _torch()
    a = np.ascontiguousarray(a)
    view = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64, np.dtype(np.uint8): np.uint8}.get(a.dtype)
    if view is not None and a.dtype != np.uint8:
        a = a.view(view)
    return torch.from_numpy(a.copy()).cuda()


def host(t, dtype):
    return t.cpu().numpy().view(dtype)


def ptr(t):
    return C.c_void_p(t.data_ptr())


def requests_model(leaves, flags, first, last, ranges):
    """extractMarkedElements per peer (R/domain/layout.hpp:104-139): runs of flagged leaves inside each peer's range as
    (first key, key behind the last) pairs; flagged leaves outside every range and outside [first, last) are unmatched"""
    pairs, counts = [], []
    owned = np.zeros(len(flags), bool)
    for a, b in ranges:
        c = 0
        while a != b:
            while a < b and flags[a] == 0:
                owned[a] = True
                a += 1
            if a != b:
                pairs.append(leaves[a])
                while a < b and flags[a] == 1:
                    owned[a] = True
                    a += 1
                pairs.append(leaves[a])
                c += 1
        counts.append(c)
    outside = np.ones(len(flags), bool)
    outside[first:last] = False
    return np.array(pairs, leaves.dtype), counts, int(np.count_nonzero((flags != 0) & ~owned & outside))


@pytest.mark.parametrize("kb", [32, 64])
def test_halo_request_rows_equal_the_host_variant_and_the_model(hip, kb):
    torch = _torch()
    rng = np.random.default_rng(5)
    kdt = np.uint32 if kb == 32 else np.uint64
    for L, P, fail in ((5000, 4, 0), (70000, 7, 0), (300, 3, 1), (4096, 2, 0)):
        top = (1 << (30 if kb == 32 else 63)) - 1
        leaves = np.sort(rng.choice(top, L + 1, replace=False).astype(kdt))
        cuts = np.sort(rng.choice(np.arange(1, L), P - 1, replace=False))
        bounds = np.concatenate([[0], cuts, [L]])
        me = int(rng.integers(0, P))
        first, last = int(bounds[me]), int(bounds[me + 1])
        flags = (rng.random(L) < 0.3).astype(np.int32)
        flags[first:last] = 0
        # the peers: every rank but me and (if there are more than two) one other, whose flagged leaves are unmatched
        skip = (me + 1) % P if P > 2 else -1
        ranges = np.zeros((P, 2), np.int32)
        for r in range(P):
            if r != me and r != skip:
                ranges[r] = bounds[r], bounds[r + 1]
        want_pairs, want_counts, want_bad = requests_model(leaves, flags, first, last, [tuple(r) for r in ranges])

        dl, df = dev(leaves), dev(flags)
        pairs_a = torch.zeros(L + 2, dtype=dl.dtype, device="cuda")
        counts = (C.c_uint32 * P)()
        bad = C.c_uint32(77)
        rr = (C.c_int32 * (2 * P))(*[int(v) for v in ranges.ravel()])
        hip._chk(hip.lib.cstone_hip_halo_requests(hip.h, C.c_int(kb), ptr(dl), ptr(df), C.c_int(L), C.c_int(first),
                                                  C.c_int(last), rr, C.c_int(P), ptr(pairs_a), counts, C.byref(bad)),
                 "halo_requests")
        assert list(counts) == want_counts and bad.value == want_bad
        assert np.array_equal(host(pairs_a, kdt)[:want_pairs.size], want_pairs)

        pairs_b = torch.zeros(L + 2, dtype=dl.dtype, device="cuda")
        row = torch.full((P + 1,), -1, dtype=torch.int64, device="cuda")
        hip._chk(hip.lib.cstone_hip_halo_request_rows(hip.h, C.c_int(kb), ptr(dl), ptr(df), C.c_int(L), C.c_int(first),
                                                      C.c_int(last), rr, C.c_int(P), ptr(pairs_b), ptr(row), C.c_int(fail)),
                 "halo_request_rows")
        hip.sync()
        got = host(row, np.uint64)
        assert [int(v) for v in got[:P]] == [2 * c for c in want_counts]  # keys, i.e. two per pair
        assert int(got[P]) == (2 if fail else (1 if want_bad else 0))
        assert np.array_equal(host(pairs_b, kdt)[:want_pairs.size], want_pairs)


def test_peer_range_counts(hip):
    """the treelet sizes of syncTreelets from the search results of translateAssignment (exchange_focus.hpp:61-96):
    row[p] = leaves over rank p's range + 1 for a peer, 0 otherwise"""
    torch = _torch()
    rng = np.random.default_rng(6)
    for P in (1, 2, 5, 64, 300):
        above = np.sort(rng.integers(0, 10**6, P + 1)).astype(np.uint64)          # findNodeAbove(assignment[r])
        below_plus = above + rng.integers(0, 3, P + 1).astype(np.uint64)           # first leaf >= assignment[r] + 1
        bounds = np.concatenate([above, below_plus])
        peer = (rng.random(P) < 0.5).astype(np.uint8)
        want = []
        for p in range(P):
            s, e = int(above[p]), int(below_plus[p + 1]) - 1
            e = max(e, s)
            want.append((e - s) + 1 if peer[p] else 0)
        row = torch.full((P,), -1, dtype=torch.int64, device="cuda")
        db = dev(bounds)
        hip._chk(hip.lib.cstone_hip_peer_range_counts(hip.h, ptr(db), (C.c_uint8 * P)(*peer.tolist()), C.c_int(P),
                                                      ptr(row)), "peer_range_counts")
        hip.sync()
        assert [int(v) for v in host(row, np.uint64)] == want


def test_add_macs_and_adjacent_difference(hip):
    torch = _torch()
    rng = np.random.default_rng(8)
    for L in (1, 63, 1000, 123457):
        M = L + (L - 1) // 7
        macs = (rng.random(M) < 0.2).astype(np.int8)
        lti = rng.permutation(M)[:L].astype(np.int32)  # node of every leaf
        flags = (rng.random(L) < 0.1).astype(np.int32)
        df, dm, dl = dev(flags), torch.from_numpy(macs).cuda(), dev(lti)  # (named: a temporary's block would be reused)
        hip._chk(hip.lib.cstone_hip_add_macs(hip.h, ptr(dm), ptr(dl), C.c_int(L), ptr(df)), "add_macs")
        want = np.where(macs[lti] != 0, 1, flags)  # FocusedOctree::addMacs: marked leaves become halo candidates
        assert np.array_equal(df.cpu().numpy(), want)
        offs = np.concatenate([[0], np.cumsum(rng.integers(0, 50, L))]).astype(np.uint32)
        out = torch.zeros(L, dtype=torch.int32, device="cuda")
        do = dev(offs)
        hip._chk(hip.lib.cstone_hip_adjacent_difference_u32(hip.h, ptr(do), C.c_size_t(L), ptr(out)),
                 "adjacent_difference")
        assert np.array_equal(host(out, np.uint32), np.diff(offs))
