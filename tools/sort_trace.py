#!/usr/bin/env python3
"""phase breakdown of the onesweep pass from a CSTONE_SORT_TRACE build (tools/build_variant.sh trace -DCSTONE_SORT_TRACE):
   CSTONE_HIP_LIB=.../variants/trace.so CSTONE_SORT_TRACE_FILE=/tmp/t.bin python tools/sort_bench.py --reps 1
   python tools/sort_trace.py /tmp/t.bin"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
P, rows, slots, tile = [int(v) for v in raw[:4]]
t = raw[4:].reshape(P, rows, slots).astype(np.int64)
tick_ns = 10.0  # wall_clock64: 100 MHz
names = ["entry->ticket", "ticket->keys", "keys->ranked", "rank barrier", "digit totals", "wave offsets",
         "permute keys", "look-back", "lb barrier", "store keys", "vals->lds", "store vals"]
for p in range(P):
    tp = t[p]
    ok = (tp[:, :13] > 0).all(axis=1)
    tp = tp[ok]
    if tp.shape[0] == 0:
        continue
    d = np.diff(tp[:, :13], axis=1) * tick_ns / 1e3  # us
    span = (tp[:, 12].max() - tp[:, 0].min()) * tick_ns / 1e3
    life = (tp[:, 12] - tp[:, 0]) * tick_ns / 1e3
    print(f"pass {p}: {tp.shape[0]} tiles, kernel span {span:.1f} us, tile lifetime mean {life.mean():.2f} us "
          f"(p50 {np.median(life):.2f}, p95 {np.percentile(life, 95):.2f}); look-back rounds mean {(tp[:,13] & 0xFFFF).mean():.2f} "
          f"(empty {((tp[:,13] >> 16) & 0xFFFF).mean():.2f}, partial {((tp[:,13] >> 32) & 0xFFFF).mean():.2f}) "
          f"rows mean {tp[:,14].mean():.2f} max {tp[:,14].max()}")
    print("   median us: " + "  ".join(f"{n} {v:.2f}" for n, v in zip(names, np.median(d, axis=0))))
    print("   mean   us: " + "  ".join(f"{n} {v:.2f}" for n, v in zip(names, d.mean(axis=0))))
    # concurrency: average number of tiles alive
    print(f"   mean tiles alive {life.sum() / span:.1f}")
    if p == 0:
        xcc = (tp[:, 15] >> 32) & 0xF
        print("   tiles per XCC:", np.bincount(xcc.astype(int)).tolist())
