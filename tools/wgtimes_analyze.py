#!/usr/bin/env python3
"""Reads the raw table HPRLP_WG_TIMES=1 HPRLP_WG_TIMES_DUMP=<file> writes at solver destruction (one block per tiled
matrix: the LAST launch of k_tiled_fused on it) and prints where the launch's time goes: ramp-up, per-super-block
durations by round, end-time spread per XCD and per CU, idle tail."""
import collections
import sys

import numpy as np

blocks, cur = [], None
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        cur = []
        blocks.append(cur)
        continue
    cur.append([float(x) for x in ln.split()])
for bi, b in enumerate(blocks):
    a = np.array(b)
    if a.size == 0:
        continue
    wg, xcc, cu, start, ends, end = a[:, 0].astype(int), a[:, 1].astype(int), a[:, 2].astype(int), a[:, 3], a[:, 4:9], a[:, 9]
    span = end.max()
    print(f"== matrix {bi}: {len(a)} workgroups, span {span:.1f} us; start {start.min():.1f}..{start.max():.1f}; "
          f"end min {end.min():.1f} mean {end.mean():.1f} max {span:.1f}; idle tail {100 * (1 - end.mean() / span):.1f} % of workgroup-time")
    prev = start
    for q in range(5):
        ok = ends[:, q] >= 0
        if not ok.any():
            break
        d = (ends[:, q] - prev)[ok]
        print(f"   super-block {q + 1} of a workgroup: n={ok.sum():4d}  duration min {d.min():6.1f} p10 {np.percentile(d, 10):6.1f} "
              f"mean {d.mean():6.1f} p90 {np.percentile(d, 90):6.1f} max {d.max():6.1f} us")
        prev = np.where(ok, ends[:, q], prev)
    nsb = (ends >= 0).sum(axis=1)
    for k in sorted(set(nsb)):
        e = end[nsb == k]
        print(f"   workgroups with {k} super-blocks: n={len(e)}, end {e.min():.1f}..{e.max():.1f} (mean {e.mean():.1f})")
    print("   per XCD (id from blockIdx % 8 | XCC_ID seen): end mean / max, super-blocks done")
    for x in range(8):
        sel = wg % 8 == x
        print(f"     XCD {x}: xcc ids {sorted(set(xcc[sel]))}  end mean {end[sel].mean():6.1f} max {end[sel].max():6.1f}  super-blocks {nsb[sel].sum()}")
    # per CU: the workgroups that shared it
    bycu = collections.defaultdict(list)
    for i in range(len(a)):
        bycu[(xcc[i], cu[i])].append(i)
    sizes = collections.Counter(len(v) for v in bycu.values())
    cu_end = np.array([end[v].max() for v in bycu.values()])
    cu_work = np.array([nsb[v].sum() for v in bycu.values()])
    print(f"   CUs seen {len(bycu)}, workgroups per CU {dict(sizes)}; CU end time min {cu_end.min():.1f} mean {cu_end.mean():.1f} max {cu_end.max():.1f}")
    for k in sorted(set(cu_work)):
        e = cu_end[cu_work == k]
        print(f"     CUs with {k} super-blocks in total: n={len(e)}, end {e.min():.1f}..{e.max():.1f} (mean {e.mean():.1f}); rate {1e0 * k / e.mean() * 1e3:.2f} super-blocks/ms")
    pair = [abs(end[v[0]] - end[v[1]]) for v in bycu.values() if len(v) == 2 and nsb[v[0]] == nsb[v[1]]]
    if pair:
        print(f"   |end difference| of the two workgroups of a CU (equal counts): mean {np.mean(pair):.1f} max {np.max(pair):.1f} us")
