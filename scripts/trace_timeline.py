#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace (+ --memory-copy-trace) run: per time bucket, the fraction of the bucket in which at
least one dispatch of each kernel group was running, and the copies.  usage: trace_timeline.py <rocprof dir> [bucket_ms] > table"""
import csv, glob, sys, collections, re

d = sys.argv[1]
bucket = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 50e6
rows = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("void ", "")
        if "(" in k and k.index("(") > 0: k = k[:k.index("(")]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
for fn in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")))
if not rows:
    sys.exit("no trace rows under " + d)
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
def group(k):
    for pat, g in (("enc5_walk_kernel<0", "count"), ("enc5_walk_kernel<1", "emit"), ("enc5_walk_kernel<2", "gather"), ("enc5_fold", "fold"), ("enc5_write", "write"),
                   ("enc5_", "enc5misc"), ("decode_v", "DEC"), ("encode_v3", "ENC3"), ("huffman_decode", "hufdec"), ("huffman_par", "hufpar"),
                   ("huffman_encode", "hufenc"), ("huffman_prog", "hufprog"), ("scan_check", "check"), ("copy:", None)):
        if pat in k:
            return g or k.replace("MEMORY_COPY_", "").replace("copy:", "cp")
    return "other"
names = collections.OrderedDict()
tot = collections.Counter(); cnt = collections.Counter()
nb = int((t1 - t0) / bucket) + 1
busy = collections.defaultdict(lambda: [0.0] * nb)
by = collections.defaultdict(list)
for s, e, k in rows:
    g = group(k); by[g].append((s, e)); tot[k] += e - s; cnt[k] += 1
for g, iv in by.items():
    iv.sort()
    cs, ce = iv[0]
    merged = []
    for s, e in iv[1:]:
        if s <= ce: ce = max(ce, e)
        else: merged.append((cs, ce)); cs, ce = s, e
    merged.append((cs, ce))
    for s, e in merged:
        b = int((s - t0) / bucket)
        while s < e:
            lim = t0 + (b + 1) * bucket
            x = min(e, lim)
            busy[g][b] += (x - s) / bucket
            s = x; b += 1
cols = sorted(busy, key=lambda g: -sum(busy[g]))
print("span %.3f s, %d dispatches/copies; bucket %.0f ms; cell = tenths of the bucket in which the group had something running" % ((t1 - t0) / 1e9, len(rows), bucket / 1e6))
print("t_ms   " + " ".join("%8s" % c[:8] for c in cols))
for b in range(nb):
    if all(busy[g][b] < 0.005 for g in cols): continue
    print("%6d " % (b * bucket / 1e6) + " ".join("%8s" % ("." if busy[g][b] < 0.005 else "%.2f" % busy[g][b]) for g in cols))
print()
print("runs (consecutive dispatches of one kernel or copy kind, gaps < 2 ms merged): start ms, end ms, busy ms, n, what")
ev = sorted(rows)
runs = []
for st, en, k in ev:
    k2 = k if not k.startswith("copy:") else k
    for r in reversed(runs[-6:]):
        if r[4] == k2 and st - r[1] < 2e6:
            r[1] = max(r[1], en); r[2] += en - st; r[3] += 1
            break
    else:
        runs.append([st, en, en - st, 1, k2])
for r in runs:
    print("%9.1f %9.1f %8.1f %6d  %s" % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, r[2] / 1e6, r[3], r[4][:100]))
print()
print("per kernel: total ms, calls")
for k, v in tot.most_common(40):
    print("%10.1f %6d  %s" % (v / 1e6, cnt[k], k[:110]))
