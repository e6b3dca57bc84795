"""Compare layer tables: python tools/sweeps/lt_diff.py base_a.txt variant.txt base_b.txt [filter]"""
import re, sys


def load(f):
    d = {}
    for l in open(f):
        m = re.match(r'(\S+)\s+(\d+) (\(.*?\))\s+(\d+)\s+([\d.]+)\s+([\d.]+)', l)
        if m:
            d[(re.sub(r'<(FWD|DGRAD),', r'<\1,', m.group(1)), m.group(3))] = (int(m.group(4)), float(m.group(5)), float(m.group(6)), m.group(2))
    return d


def key_shape(d):
    # kernel names may differ between variants: key on (role prefix, shape)
    out = {}
    for (k, shp), v in d.items():
        role = 'W' if 'wgrad' in k else ('D' if 'DGRAD' in k else 'F')
        out.setdefault((role, shp), []).append((k, v))
    return out


a, b, c = (key_shape(load(f)) for f in sys.argv[1:4])
flt = sys.argv[4] if len(sys.argv) > 4 else ''
ta = tb = tc = 0
rows = []
for k in a:
    if k in b and k in c and len(a[k]) == 1 and len(b[k]) == 1 and len(c[k]) == 1:
        (ka, va), (kb, vb), (kc, vc) = a[k][0], b[k][0], c[k][0]
        n = va[0]
        ta += n * (va[1] + va[2]); tb += n * (vb[1] + vb[2]); tc += n * (vc[1] + vc[2])
        if flt in ka or flt in kb:
            rows.append((n * ((va[1] + va[2] + vc[1] + vc[2]) / 2 - vb[1] - vb[2]), ka, kb, k[1], va, vb, vc))
print('total us per step: base_a %.0f variant %.0f base_b %.0f' % (ta, tb, tc))
rows.sort(reverse=True, key=lambda r: r[0])
for r in rows[:16] + [None] + rows[-8:]:
    if r is None:
        print('...'); continue
    print('%7.1f %-40s %-40s %-36s %6.1f+%4.1f(%s) %6.1f+%4.1f(%s) %6.1f+%4.1f x%d' % (
        r[0], r[1], r[2] if r[2] != r[1] else '=', r[3], r[4][1], r[4][2], r[4][3], r[5][1], r[5][2], r[5][3], r[6][1], r[6][2], r[4][0]))
