#!/usr/bin/env python3
"""Per-setting summary of a tools/sched_ab.py run (jsonl, optionally with a text prefix per line)."""
import json, collections, sys
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    if '{' not in l: continue
    pre, js = l.split('{', 1); j = json.loads('{' + js)
    d[pre + json.dumps(j['tuning'])].append(j['ms'])
    if not j['ok']: print("WRONG PROOF:", l.strip())
for k, v in d.items(): print(k, ' '.join('%.2f' % x for x in v), ' mean %.3f min %.3f' % (sum(v) / len(v), min(v)))
