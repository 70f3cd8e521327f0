"""Fits the AUTO form rule of csrc/capi.hip to a sweep of both kernel forms (tools/form_sweep.py dense) and writes it as a
table (csrc/form_table.inc); given a second, held-out sweep, reports what AUTO would have cost there.

    python tools/form_fit.py profiles/r04_form_sweep_dense.txt [held-out sweep] > csrc/form_table.inc   (table on stdout, report on stderr)

The model.  w = wavefronts per SIMD the one-thread-per-replica form would launch.
  thread form:  rate(w) = A(k) * w / k,  k = ceil(w)     - a SIMD with k resident waves runs at A(k); with w < k on average,
                the launch lasts as long as its fullest SIMDs, so only w / k of the slots do work (the saw-tooth of
                profiles/r02_form_sweep.txt: 81 920 chains at w = 1.25 run at A(2) * 0.625, not at A(1)).  A(1..4) are the
                measured rates at w = 1, 2, 3, 4 (for the register widths that hold 2 waves per SIMD, A(3) and A(4) include
                the second round).
  lane-split:   four times the waves, a quarter of the work each: the same saw-tooth on a four times finer scale,
                rate(w) = Q(kq / 4) * 4 w / kq,  kq = ceil(4 w), with Q read off the measured curve (the sweep's w values are
                multiples of 0.25 up to 2, i.e. whole numbers of lane-split waves per SIMD; linear in between beyond).
Both are stored relative to A(4) per (dim, temperatures).  A dim with kernels of its own (dim compiled in: 20, 30, 50) uses its
own row; every other dim runs the run-time-dim kernels and is interpolated linearly between the generic grid dims of its
lane-width class (compiled-in kernels are 10-40 % faster per dimension than their generic neighbours, in both forms, so
the two families cannot share rows); between grid ladder lengths the table is interpolated in log2 T.  AUTO picks the lane-split form where Q(w) > rate(w)."""
import collections
import math
import sys


def load(path):
    tab = collections.defaultdict(dict)
    for l in open(path):
        r = l.split()
        if len(r) >= 7 and r[0].isdigit():
            tab[(int(r[0]), int(r[1]))][float(r[3])] = (float(r[4]), float(r[5]))
    return tab


def fit(tab):
    dims = sorted({d for d, _ in tab})
    temps = sorted({t for _, t in tab})
    ws = sorted(next(iter(tab.values())))
    A, Q = {}, {}
    for key, rows in tab.items():
        a4 = rows[4.0][0]
        A[key] = [rows[float(k)][0] / a4 for k in (1, 2, 3, 4)]
        Q[key] = [rows[w][1] / a4 for w in ws]
    return dims, temps, ws, A, Q


def lerp(a, b, t):
    return [x + (y - x) * t for x, y in zip(a, b)]


EXACT_DIMS = (2, 3, 4, 5, 10, 20, 30, 50)  # dims with kernels of their own (dim compiled in: csrc/variants.h), both forms


def params(model, dim, T):
    dims, temps, ws, A, Q = model
    if dim in EXACT_DIMS and dim in dims:
        cls = [dim]  # its own kernels: its own row of the table
    else:  # a run-time-dim kernel: between the generic grid dims of the same lane-width class
        cls = [d for d in dims if (d <= 32) == (dim <= 32) and d not in EXACT_DIMS]
    d0 = max([d for d in cls if d <= dim], default=cls[0])
    d1 = min([d for d in cls if d >= dim], default=cls[-1])
    td = 0.0 if d0 == d1 else (dim - d0) / (d1 - d0)
    t0 = max([t for t in temps if t <= T], default=temps[0])
    t1 = min([t for t in temps if t >= T], default=temps[-1])
    tt = 0.0 if t0 == t1 else (math.log2(T) - math.log2(t0)) / (math.log2(t1) - math.log2(t0))
    out = []
    for P in (A, Q):
        lo, hi = lerp(P[(d0, t0)], P[(d1, t0)], td), lerp(P[(d0, t1)], P[(d1, t1)], td)
        out.append(lerp(lo, hi, tt))
    return out


def quad_faster(model, dim, T, w):
    thread, quad = rates(model, dim, T, w)
    return quad > thread


def rates(model, dim, T, w):
    """(thread form, lane-split form) modelled rates relative to the thread form at 4 waves per SIMD."""
    ws = model[2]
    if w > ws[-1]:
        return 1.0, 0.0
    a, q = params(model, dim, T)
    k = max(1, math.ceil(w - 1e-9))
    thread = a[k - 1] * w / k
    # the lane-split form launches four times the waves: the same saw-tooth on a four times finer scale - the rate of the
    # next whole number of lane-split waves per SIMD (read off the measured curve), times the fill of that last wave slot
    # (ladders of more than 16 temperatures: a whole workgroup of 2-8 waves per ladder is the unit and the dispatcher spreads
    # them over the CUs: no saw-tooth of its own, the measured curve is interpolated as it is)
    kq = max(1, math.ceil(4 * w - 1e-9)) if T <= 16 else 4 * w
    wu = kq / 4.0
    if wu <= ws[0]:
        b = q[0]
    else:
        i = min(max(j for j in range(len(ws)) if ws[j] <= wu + 1e-9), len(ws) - 2)
        b = q[i] + (q[i + 1] - q[i]) * (min(wu, ws[-1]) - ws[i]) / (ws[i + 1] - ws[i])
    quad = b * (4 * w / kq)
    return thread, quad


def regret(model, tab, tag):
    worst, n, bad = 1.0, 0, []
    for (dim, T), rows in sorted(tab.items()):
        for w, (th, qu) in sorted(rows.items()):
            pick = qu if quad_faster(model, dim, T, w) else th
            r = pick / max(th, qu)
            n += 1
            worst = min(worst, r)
            if r < 0.95:
                bad.append((dim, T, w, round(r, 3)))
    print(f"{tag}: {n} cases, worst AUTO / best pinned form {worst:.3f}, {len(bad)} case(s) below 0.95 {bad[:8]}", file=sys.stderr)
    return worst


def main():
    tab = load(sys.argv[1])
    model = fit(tab)
    dims, temps, ws, A, Q = model
    regret(model, tab, "fit data")
    if len(sys.argv) > 2:
        regret(model, load(sys.argv[2]), "held-out data")
    f = lambda v: ", ".join(f"{x:.4f}f" for x in v)  # noqa: E731
    print(f"// generated by tools/form_fit.py from {sys.argv[1]} - do not edit; see the tool for the model")
    stamp = next((l.split()[2] for l in open(sys.argv[1]) if l.startswith("# source_hash ")), None)
    print("// the kernel sources the sweep was measured on (tools/source_hash.py; ptrwm_form_table_source_hash())")
    print(f'constexpr char kFormTableSourceHash[] = "{stamp or "unknown (the sweep carries no source_hash line)"}";')
    print(f"constexpr int kFormDims[] = {{{', '.join(map(str, dims))}}};")
    print(f"constexpr bool kFormDimExact[] = {{{', '.join('true' if d in EXACT_DIMS else 'false' for d in dims)}}};  // a kernel with this dim compiled in")
    print(f"constexpr int kFormTemps[] = {{{', '.join(map(str, temps))}}};")
    print(f"constexpr float kFormW[] = {{{f(ws)}}};")
    print(f"constexpr int kFormND = {len(dims)}, kFormNT = {len(temps)}, kFormNW = {len(ws)};")
    print("// thread form: rate at w = 1, 2, 3, 4 waves per SIMD relative to w = 4")
    print("constexpr float kFormThread[kFormND][kFormNT][4] = {")
    for d in dims:
        print("  {" + ", ".join("{" + f(A[(d, t)]) + "}" for t in temps) + "},")
    print("};\n// lane-split form: rate at the w of kFormW relative to the thread form at w = 4")
    print("constexpr float kFormQuad[kFormND][kFormNT][kFormNW] = {")
    for d in dims:
        print("  {" + ",\n   ".join("{" + f(Q[(d, t)]) + "}" for t in temps) + "},")
    print("};")


if __name__ == "__main__":
    main()
