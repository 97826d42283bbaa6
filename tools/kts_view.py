#!/usr/bin/env python3
"""View an in-kernel timestamp trace (WM_TRACE_EVENTS=<file>): find a window where two passes decode together and print
one step's worth of events for both, in µs.  kinds: 0 cross-attention start, 1 merge start (= cross-attention end),
2 logits start, 3 argmax start."""
import sys, collections
ev = [tuple(map(int, l.split())) for l in open(sys.argv[1])]
ev.sort(key=lambda e: e[2])
t0 = ev[0][2]
US = 100.0  # ticks per µs (100 MHz)
# windows with two owners active: look at consecutive argmax (kind 3) events
am = [(e[2], e[0]) for e in ev if e[1] == 3]
# find a stretch of >= 40 argmax events alternating between two ids with gaps < 600 µs
best = None
for i in range(len(am) - 40):
    seg = am[i:i + 40]
    if len({o for _, o in seg}) == 2 and all((b[0] - a[0]) / US < 600 for a, b in zip(seg, seg[1:])):
        best = i + 10
        break
if best is None:
    print("no two-pass window found"); sys.exit(0)
ta = am[best][0]; tb = am[best + 6][0]
names = {0: "X start", 1: "X end/merge", 2: "logits", 3: "argmax"}
last = {}
print(f"window {((ta - t0) / US / 1e3):.2f} .. {((tb - t0) / US / 1e3):.2f} ms")
for o, k, t in ev:
    if ta <= t <= tb:
        d = (t - last.get(o, t)) / US
        last[o] = t
        print(f"{(t - ta) / US:9.1f} us  pass {o}  {'    ' * (o - 1)}{names[k]:12s} (+{d:.1f})")
# per-owner stats inside two-pass stretches: X duration and L gap (merge start -> next X start)
xs, ls = collections.defaultdict(list), collections.defaultdict(list)
prev = {}
for o, k, t in ev:
    if not (am[best][0] <= t <= am[min(len(am) - 1, best + 60)][0]): continue
    if k == 1 and prev.get(o, (None,))[0] == 0: xs[o].append((t - prev[o][1]) / US)
    if k == 0 and prev.get(o, (None,))[0] == 1: ls[o].append((t - prev[o][1]) / US)
    prev[o] = (k, t)
for o in xs:
    print(f"pass {o}: X duration avg {sum(xs[o])/len(xs[o]):.1f} us (n={len(xs[o])}); merge->next X (same step) avg {sum(ls[o])/max(1,len(ls[o])):.1f} us")
