#!/usr/bin/env python3
"""Offline: what the packets of db_group_union_kernel took (TKNN_DB_DUMP files of the diagnostic library, one per pass:
TKNN_DB_DIAG=512+4096 first pass, 512+2048 second) -- percentiles, what list order / longest-first end at on `waves` waves,
and how well extent, rounds and settles predict a packet's time.   scripts/db_packet_stats.py pass1.bin pass2.bin [waves]"""
import heapq
import sys

import numpy as np


def sched(times, order, m):
    h = [0.0] * m
    heapq.heapify(h)
    for i in order:
        heapq.heappush(h, heapq.heappop(h) + times[i])
    return max(h)


def main():
    files = [a for a in sys.argv[1:] if not a.isdigit()]
    waves = int(next((a for a in sys.argv[1:] if a.isdigit()), 4608))
    recs = [np.fromfile(f, dtype=np.int32).reshape(-1, 4) for f in files]
    for f, p in zip(files, recs):
        t = p[:, 0].astype(float)
        print("%s: %d packets, mean per wave %.0f, percentiles 5/25/50/75/95/99/100 %s" % (f, len(t), t.sum() / waves, np.percentile(t, [5, 25, 50, 75, 95, 99, 100]).astype(int)))
        print("   list order ends at %.0f, longest first at %.0f" % (sched(t, range(len(t)), waves), sched(t, np.argsort(-t), waves)))
        for j, lab in ((1, "extent"), (2, "rounds"), (3, "settles")):
            print("   correlation with %-7s %.2f; longest-%s-first ends at %.0f" % (lab, np.corrcoef(t, p[:, j])[0, 1], lab, sched(t, np.argsort(-p[:, j]), waves)))
    if len(recs) == 2 and len(recs[0]) == len(recs[1]):
        a, b = recs
        print("first pass against second: ticks %.2f, rounds %.2f" % (np.corrcoef(a[:, 0], b[:, 0])[0, 1], np.corrcoef(a[:, 2], b[:, 2])[0, 1]))


if __name__ == "__main__":
    main()
