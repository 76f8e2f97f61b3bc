"""Single-fault explanations for the wrong packed-fp32 results recorded by profiles/ub/pk_hazard (second argument = records file).

A record holds the operands of the victim's chain  r_c = fma(x3_c, w.w, fma(x2_c, w.z, fma(x0_c, w.x, x1_c * w.y)))  for the four float4
components, the value the packed instructions produced and the value the scalar instructions produced.  For every wrong component this
script looks for ONE substitution that reproduces the wrong bits: a step skipped (its destination kept the old value), an operand taken
from another component (the other half of the register pair / the other pair), a weight taken from another slot, or the result of
another component.
"""
import re
import sys
from collections import Counter

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))     # a*b exact in double; the double rounding of the sum is negligible here


def chain(x, w, c, sub=None):
    """x[k][c]; steps: 0 mul(x1, wy)  1 fma(x0, wx)  2 fma(x2, wz)  3 fma(x3, ww).  sub = (step, kind, arg)."""
    order = [(1, 1), (0, 0), (2, 2), (3, 3)]      # (operand row, weight slot) per step
    acc = f32(0)
    for step, (k, wi) in enumerate(order):
        xv, wv = f32(x[k][c]), f32(w[wi])
        if sub and sub[0] == step:
            if sub[1] == "skip":
                continue
            if sub[1] == "xcomp":
                xv = f32(x[k][sub[2]])
            if sub[1] == "wslot":
                wv = f32(w[sub[2]])
            if sub[1] == "src2zero":
                acc = f32(0)
        acc = f32(xv * wv) if step == 0 else fma(xv, wv, acc)
    return acc


def explain(rec):
    out = []
    x, w = rec["x"], rec["w"]
    for c in range(4):
        got, want = f32(rec["got"][c]), f32(rec["want"][c])
        if got.tobytes() == want.tobytes():
            continue
        assert chain(x, w, c).tobytes() == want.tobytes(), "host re-evaluation of the chain disagrees with the kernel's scalar result"
        found = []
        for c2 in range(4):
            if c2 != c and f32(rec["want"][c2]).tobytes() == got.tobytes():
                found.append(f"= the correct result of component {c2}")
        for step in range(4):
            for kind, args in (("skip", [None]), ("src2zero", [None]), ("xcomp", [a for a in range(4) if a != c]), ("wslot", range(4))):
                for a in args:
                    if chain(x, w, c, (step, kind, a)).tobytes() == got.tobytes():
                        found.append(f"step {step} {kind}{'' if a is None else ' ' + str(a)}")
        out.append((c, found))
    return out


def main():
    tally = Counter()
    n = 0
    # the records are written with C's %a (hex floats), which JSON cannot carry: parse them by hand
    hexf = re.compile(r"-?0x[0-9a-f.]+p[-+]?\d+|-?inf|-?nan")
    for line in open(sys.argv[1]):
        nums = [float.fromhex(t) if t.lstrip("-").startswith("0x") else float(t) for t in hexf.findall(line)]
        meta = dict(re.findall(r'"(agg|fill|lane|block|iter)": ("[^"]*"|\d+)', line))
        w, xs, got, want = nums[0:4], nums[4:20], nums[20:24], nums[24:28]
        rec = {"w": w, "x": [xs[0:4], xs[4:8], xs[8:12], xs[12:16]], "got": got, "want": want}
        for c, found in explain(rec):
            n += 1
            key = found[0] if found else "no single-fault explanation"
            tally[(c, key)] += 1
            if n <= 12:
                print(f"lane {meta.get('lane')} comp {c}: got {f32(got[c])!r} want {f32(want[c])!r}: {found or 'no single-fault explanation'}   [{meta.get('agg')}]")
    print(f"\n{n} wrong components analysed")
    for (c, key), k in sorted(tally.items(), key=lambda kv: -kv[1]):
        print(f"  component {c}: {k:5d} x {key}")


if __name__ == "__main__":
    main()
