"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py: keeps only the kernels of the
last `--steps` optimizer steps (delimited by the fused-Adam launches), so MIOpen's find-mode
benchmarking during warm-up does not pollute the numbers.  Usage: step_profile.py <kernel_trace.csv>"""
import collections, csv, sys

def cat(n):
    if 'dwconv' in n: return 'ppea dwconv'
    if '::bn_' in n: return 'ppea bn_fused'
    if any(k in n for k in ('conv_nhwc', 'conv_wgrad', 'conv_pack', 'image_to_nhwc', 'conv_image', 'image_pack', 'image_wgrad')):
        return 'ppea dense conv'
    if 'CatArray' in n: return 'cat'
    if any(k in n for k in ('ssim', 'backproject', 'grid_sample', 'smooth_', 'loss_select', 'cost_volume', 'pack_filter', 'pwconv', 'conv3x3', 'adam_')): return 'ppea other'
    if n.startswith('Cijk'): return 'rocBLAS/hipBLASLt GEMM'
    if 'igemm' in n.lower(): return 'MIOpen igemm'
    if 'Col2Im' in n or 'Im2d2Col' in n or 'Im2Col' in n: return 'im2col/col2im'
    if 'batch_norm' in n or 'BatchNorm' in n: return 'batch_norm (ATen/MIOpen)'
    if 'reflection_pad' in n: return 'reflection_pad'
    if 'multi_tensor' in n: return 'multi_tensor(adam/foreach)'
    if 'elementwise' in n: return 'elementwise'
    if 'reduce_kernel' in n: return 'reduce'
    if 'transpose' in n or 'SubTensor' in n or ('Op' in n and 'Tensor' in n): return 'MIOpen tensor ops'
    if 'ck::' in n or '_ZN2ck' in n or 'conv' in n.lower() or 'gridwise' in n: return 'MIOpen/CK conv'
    return 'other'

def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if 'FusedOptimizer' in r[2] or 'fused_adam' in r[2].lower() or 'adam_' in r[2]]
    # group consecutive adam launches into step ends
    ends = []
    for i in adam:
        if not ends or rows[i][0] - rows[ends[-1]][1] > 20e6:
            ends.append(i)
        else:
            ends[-1] = i
    if len(ends) < steps + 1:
        print("not enough steps found", len(ends)); return
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    wall = (sel[-1][1] - sel[0][0]) / 1e6 / steps
    agg = collections.defaultdict(lambda: [0.0, 0])
    per = collections.defaultdict(lambda: [0.0, 0])
    for s, e, n, *_ in sel:
        agg[cat(n)][0] += (e - s); agg[cat(n)][1] += 1
        per[n][0] += (e - s); per[n][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"steps {steps}: wall {wall:.1f} ms/step, kernel time {tot/1e6/steps:.1f} ms/step, {len(sel)//steps} launches/step")
    for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
        print(f"  {k:30s} {t/1e6/steps:8.2f} ms/step {100*t/tot:5.1f}%  {c//steps:6d} launches/step")
    # concurrency: share of the step's wall time with 0 / 1 / 2 / 3+ kernels in flight (parallel graph branches)
    ev = []
    for st, en, *_ in sel:
        ev.append((st, 1)); ev.append((en, -1))
    ev.sort()
    hist, depth, last = collections.defaultdict(float), 0, ev[0][0]
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        depth += d; last = t
    span = sum(hist.values())
    print("kernels in flight: " + "  ".join(f"{k if k < 3 else '3+'}: {100 * v / span:.1f}%" for k, v in sorted(hist.items())))
    # per HSA queue / HIP stream: the graph's parallel branches land on different queues; the busiest one bounds the step
    for col, name in ((3, 'queue'), (4, 'stream')):
        q = collections.defaultdict(lambda: [0.0, 0, collections.defaultdict(float)])
        for r in sel:
            q[r[col]][0] += r[1] - r[0]; q[r[col]][1] += 1; q[r[col]][2][cat(r[2])] += r[1] - r[0]
        for k, (t, c, cats) in sorted(q.items(), key=lambda x: -x[1][0]):
            tops = ", ".join(f"{a} {b/1e6/steps:.1f}" for a, b in sorted(cats.items(), key=lambda x: -x[1])[:5])
            print(f"  {name} {k}: {t/1e6/steps:7.2f} ms/step {c//steps:5d} launches/step  [{tops}]")
    print("top kernels:")
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    for n, (t, c) in sorted(per.items(), key=lambda x: -x[1][0])[:top]:
        print(f"  {t/1e6/steps:7.2f} ms/step {c//steps:5d}/step avg {t/c/1e3:8.1f} us  {n[:150]}")

main()
