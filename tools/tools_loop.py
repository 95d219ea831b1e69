#!/usr/bin/env python3
"""Print instruction mix of the innermost loops of a kernel (by mangled-name substring) from the -save-temps ISA."""
import collections, re, subprocess, sys, os
os.makedirs('/tmp/asm', exist_ok=True)
s = ""
for unit in ("pyrovi", "f64", "lean"):        # the three translation units of the library
    subprocess.run("cd /tmp/asm && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -save-temps -c /root/repo/pyro_amd/csrc/%s.hip -o /tmp/asm/%s.o 2>/dev/null" % (unit, unit), shell=True)
    s += open('/tmp/asm/%s-hip-amdgcn-amd-amdhsa-gfx950.s' % unit).read()
for name in sys.argv[1:]:
    show = name.endswith('+')
    name = name.rstrip('+')
    i = s.find(name); i = s.find(':', i); j = s.find('.Lfunc_end', i)
    lines = s[i:j].splitlines()
    labels = {}
    for n, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = n
    loops = []
    for n, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.append((labels[m.group(1)], n))
    print(name, 'total lines', len(lines))
    for a, b in loops:
        body = lines[a:b + 1]
        if not any('v_floor_f32' in x for x in body): continue
        if any((a2 >= a and b2 <= b and (a2, b2) != (a, b)) for a2, b2 in loops): continue
        ops = collections.Counter()
        for l in body:
            m = re.match(r'^\s+([a-z_0-9]+)', l)
            if m:
                o = m.group(1)
                k = 'VALU' if o.startswith('v_') else 'SALU' if o.startswith('s_') and not o.startswith(('s_load', 's_waitcnt', 's_cbranch', 's_branch', 's_nop')) else '_'.join(o.split('_')[:2])
                ops[k] += 1
        print('  loop', a, b, 'floors', sum('v_floor_f32' in l for l in body), 'f64', sum('_f64' in l for l in body), dict(ops))
        if show: print("\n".join(body))
