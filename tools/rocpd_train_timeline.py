"""GPU timeline of ONE training step in a rocprofv3 rocpd result (kernel trace of tools/probes/train_host.py): the step between the
last two k_sort_verts launches, cut into phases at known kernels, with each phase's span, the time a kernel was running (union over
streams), the idle time and the number of launches; then the largest idle gaps.

    python tools/rocpd_train_timeline.py <results.db> > profiles/rNN_train_timeline.txt
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()


def short(name):
    s = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", s).replace("void ", "")[:60]


starts = [i for i, r in enumerate(rows) if "k_sort_verts" in r[0]]
i0, i1 = starts[-2], starts[-1]
step = [(short(n), s, e) for n, s, e in rows[i0:i1]]
t0 = step[0][1]
# phases: (label, predicate on kernel name) -- a phase starts at the first kernel matching after the previous phase's start
marks = [("body tables / hypernetwork / skinning query / pack", lambda k: "k_sort_verts" in k),
         ("ray tracer: loops A+B", lambda k: k.startswith("k_trace_begin") or k.startswith("k_sdf_march") or k.startswith("k_trace_finish")),
         ("ray tracer: sampling + loop C", lambda k: k.startswith("k_sample_depths")),
         ("regulariser probe, compaction, re-attachment, skin jacobian", lambda k: k.startswith("k_canon_finalize") or k.startswith("k_shade_train<")),
         ("ShadeSamples forward", lambda k: k.startswith("k_shade_train<true, false") or k.startswith("k_shade_train<false, false")),
         ("compositing + loss", lambda k: k.startswith("k_composite_train_fwd")),
         ("backward: loss .. compositing", lambda k: k.startswith("k_composite_train_bwd")),
         ("backward: ShadeSamples kernel", lambda k: k.startswith("k_shade_train<true, true") or k.startswith("k_shade_train<false, true")),
         ("backward: the rest", lambda k: False),
         ("optimizer", lambda k: "multi_tensor_apply" in k and "FusedOptimizer" in k or "fused_adam" in k.lower())]
cuts = []
pos = 0
for label, pred in marks:
    for j in range(pos, len(step)):
        if pred(step[j][0]):
            cuts.append((label, j))
            pos = j + 1
            break
# "backward: the rest" starts behind the LAST k_shade_train backward launch of the main op
bw = [j for j, k in enumerate(step) if k[0].startswith("k_shade_train<true, true") or k[0].startswith("k_shade_train<false, true")]
if bw:
    cuts.append(("backward: behind the last k_shade_train (weight gradients, skinning, hypernetwork)", bw[-1] + 1))
cuts.sort(key=lambda c: c[1])
cuts.append(("(end)", len(step)))
span_all = step[-1][2] - t0


def union(seg):
    busy, cur_s, cur_e = 0, None, None
    for _, s, e in sorted(seg, key=lambda r: r[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


print("one training step: %d kernels, span %.3f ms, a kernel running %.3f ms, idle %.3f ms" % (
    len(step), span_all / 1e6, union(step) / 1e6, (span_all - union(step)) / 1e6))
print("%-88s %8s %8s %8s %8s" % ("phase (from its first kernel to the next phase's)", "span ms", "busy ms", "idle ms", "kernels"))
for (label, a), (_, b) in zip(cuts[:-1], cuts[1:]):
    if b <= a:
        continue
    seg = step[a:b]
    end = step[b][1] if b < len(step) else step[-1][2]
    sp = end - seg[0][1]
    bu = union(seg)
    print("%-88s %8.3f %8.3f %8.3f %8d" % (label + "  [" + seg[0][0][:28] + "]", sp / 1e6, bu / 1e6, (sp - bu) / 1e6, len(seg)))
gaps = []
prev_end = step[0][2]
for j in range(1, len(step)):
    g = step[j][1] - prev_end
    if g > 0:
        gaps.append((g, step[j - 1][0], step[j][0], (step[j][1] - t0) / 1e6))
    prev_end = max(prev_end, step[j][2])
gaps.sort(reverse=True)
print("largest idle gaps (us, at ms, after kernel -> before kernel):")
for g, a, b, at in gaps[:25]:
    print("  %8.1f  @%7.3f  %-44s -> %s" % (g / 1e3, at, a[:44], b[:44]))
agg = {}
for k, s, e in step:
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
print("kernels of the step by time:")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %-60s %5d %10.1f us" % (k, n, us))
