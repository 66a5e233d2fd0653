#!/usr/bin/env python3
"""Build-time check of the hand-ordered MFMA asm blocks: hipcc pads no hazards across an inline-asm
boundary, so no VALU instruction may write a register -- a VGPR, or an accumulation register through
v_accvgpr_write / v_accvgpr_mov -- that an asm MFMA block reads within the two instructions before the
block (VALU write -> MFMA operand needs 2 wait states).  Blocks that start with their own s_nop (the GUARD
variants) are exempt.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only X.hip -o X.s && check_asm_hazards.py X.s
"""
import re
import sys


def regs(tok):
    """Registers named in an operand string, as ('v' | 'a', index) pairs (VGPRs and accumulation registers)."""
    out = set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1):
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def main(path):
    lines = [l.rstrip() for l in open(path)]
    bad = 0
    blocks = 0
    i = 0
    while i < len(lines):
        if '#ASMSTART' in lines[i]:
            j = i + 1
            body = []
            while '#ASMEND' not in lines[j]:
                body.append(lines[j].strip())
                j += 1
            mf = [b for b in body if b.startswith('v_mfma')]
            if mf and not body[0].startswith('s_nop'):
                blocks += 1
                reads = set()
                for b in mf:
                    ops = b.split(None, 1)[1].split(',')
                    for o in ops[1:]:
                        reads |= regs(o)
                # previous two real instructions
                k, prev = i - 1, []
                while k >= 0 and len(prev) < 2:
                    t = lines[k].strip()
                    if t and not t.startswith(';') and not t.startswith('.') and not t.endswith(':'):
                        prev.append(t)
                    k -= 1
                for t in prev:
                    op = t.split()[0]
                    if op.startswith('v_') and not op.startswith('v_mfma'):
                        dst = regs(t.split(None, 1)[1].split(',')[0])
                        if dst & reads:
                            bad += 1
                            print('%s:%d: VALU write %r feeds the asm MFMA block at line %d' % (path, k + 2, t, i + 1))
            i = j
        i += 1
    print('%s: %d asm MFMA blocks checked, %d hazards' % (path, blocks, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(max(main(p) for p in sys.argv[1:]))
