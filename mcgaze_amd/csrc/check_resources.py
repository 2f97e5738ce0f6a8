#!/usr/bin/env python3
"""Build gate: reads the -Rpass-analysis=kernel-resource-usage remarks of one translation unit (hipcc's stderr, saved by the Makefile as
<unit>.res.txt) and FAILS the build when a kernel that holds hand-issued loads in flight lands in scratch.

wino_x3w_kernel (wino_x3.hpp) issues its weight fragments with `asm volatile("buffer_load_dwordx4 ... =&v")` and waits for them with a
hand-placed s_waitcnt: the compiler believes the destination registers are written when the instruction issues.  That is only safe while
it never copies, spills or re-uses those registers before the wait -- with 256 accumulators + 189 VGPRs the kernel sits close to the 512
budget, and one spilled fragment would give timing-dependent wrong bits that the bit-identity test catches only when the timing exposes
it.  The 8-wave Winograd tiles (engine option wino_tile = 0 / 1 / 2) have no such loads and stay selectable at run time.
(bneck_x3_kernel<128, 1, 128, *> spills 17 - 28 registers: ordinary compiler-tracked loads, a cost and not a hazard -- listed by --table.)

usage: check_resources.py <unit>.res.txt [--table]"""
import re
import sys

# kernel-name prefix (demangled template name as it appears in the mangled symbol) -> limits
RULES = {
    'wino_x3w_kernel': dict(scratch=0, vgpr_spill=0, unit='engine'),   # unit: the translation unit that instantiates it -- the gate must SEE it there
}


def parse(path):
    kernels, cur = [], None
    for line in open(path, errors='replace'):
        m = re.search(r'remark: (?:\s*)Function Name: (\S+)', line)
        if m:
            cur = dict(name=m.group(1))
            kernels.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return kernels


def main():
    path = sys.argv[1]
    kernels = parse(path)
    other = [l.rstrip() for l in open(path, errors='replace') if 'Rpass-analysis=kernel-resource-usage' not in l and not re.match(r'\s+\d* *\|', l) and l.strip()
             and not l.startswith('In file included from') and not re.match(r'\d+ (warning|remark)s? generated', l.strip())]
    for l in other:               # real warnings of the unit stay visible
        print(l, file=sys.stderr)
    bad = []
    import os
    unit = os.path.basename(path).split('.')[0]
    for prefix, lim in RULES.items():      # a compiler that stops printing the remarks (or renames them) must not turn the gate into a no-op
        if lim.get('unit') == unit and not any(prefix in k['name'] and 'VGPRs Spill' in k and 'ScratchSize [bytes/lane]' in k for k in kernels):
            bad.append(f'{prefix}: no resource remarks found in {path} (expected -Rpass-analysis=kernel-resource-usage output for it)')
    for k in kernels:
        for prefix, lim in RULES.items():
            if prefix in k['name']:
                scratch = int(k.get('ScratchSize [bytes/lane]', '0'))
                spill = int(k.get('VGPRs Spill', '0'))
                if scratch > lim['scratch'] or spill > lim['vgpr_spill']:
                    bad.append(f"{k['name']}: scratch {scratch} B/lane, {spill} spilled VGPRs (limits {lim['scratch']}, {lim['vgpr_spill']})")
    if '--table' in sys.argv:
        print('| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spill | waves/SIMD |\n|---|---|---|---|---|---|---|')
        for k in kernels:
            print(f"| `{k['name']}` | {k.get('VGPRs')} | {k.get('AGPRs')} | {k.get('TotalSGPRs')} | {k.get('ScratchSize [bytes/lane]')} | {k.get('VGPRs Spill')} | {k.get('Occupancy [waves/SIMD]')} |")
    if bad:
        print('check_resources: kernels with hand-issued loads must not spill:\n  ' + '\n  '.join(bad), file=sys.stderr)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
