#!/usr/bin/env python3
"""ISA check for the kernels that issue loads from inline asm (the compiler does not know those loads are
asynchronous): between such a load and the s_waitcnt / ds_write statement that consumes it, no other
instruction may READ the destination registers (a compiler-made copy would copy stale data).

Rule checked per kernel: for every `buffer_load_dword[x4] vD, ... offen` that sits in an inline-asm block
(;;#ASMSTART .. ;;#ASMEND), scan forward until the first instruction inside a later asm block that mentions vD;
any instruction outside asm blocks that reads vD before that is an error."""
import re, sys

def regs(tok):
    # VGPR n -> n, AGPR n -> 1000 + n
    m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        base = 1000 if m.group(1) == 'a' else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.match(r'([va])(\d+)$', tok)
    return {(1000 if m.group(1) == 'a' else 0) + int(m.group(2))} if m else set()

def operands(line):
    body = line.split(None, 1)
    if len(body) < 2:
        return []
    return [t.strip() for t in re.split(r',\s*', body[1].split('//')[0])]

bad = 0
checked = 0
for path in sys.argv[1:]:
    lines = open(path).read().split('\n')
    in_app = False
    seq = []   # (text, in_app)
    for l in lines:
        t = l.strip()
        if 'ASMSTART' in t:
            in_app = True; continue
        if 'ASMEND' in t:
            in_app = False; continue
        if l.startswith('\t') and t and not t.startswith(('.', ';')):
            seq.append((t, in_app))
        elif t.endswith(':') and not t.startswith(('.L', ';')):
            seq.append(('@' + t, False))      # kernel boundary
    for i, (t, app) in enumerate(seq):
        if not app or not t.startswith('buffer_load_dword'):
            continue
        d = regs(operands(t)[0])
        checked += 1
        for j in range(i + 1, min(i + 4000, len(seq))):
            u, uapp = seq[j]
            if u.startswith('@'):
                break
            # registers the instruction READS: everything but the destination operand of instructions that have one
            body_txt = u.split(None, 1)[1] if ' ' in u else ''
            opcode = u.split()[0]
            has_dest = opcode.startswith(('v_', 'buffer_load', 'global_load', 'ds_read', 'flat_load', 's_')) and \
                not opcode.startswith(('v_cmp', 'v_nop'))
            first, rest = (body_txt.split(',', 1) + [''])[:2] if has_dest else ('', body_txt)

            def reg_set(txt):
                out = set()
                for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', txt):     # (also registers named in an asm comment)
                    if m.group(5) is not None:
                        out.add((1000 if m.group(4) == 'a' else 0) + int(m.group(5)))
                    else:
                        base = 1000 if m.group(1) == 'a' else 0
                        out |= set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
                return out
            written, touched = reg_set(first), reg_set(rest)
            if not ((touched | written) & d):
                continue
            if uapp:
                break            # consumed (or re-loaded) by an asm statement: fine
            if not (touched & d):
                break            # only redefined by the compiler (a textually later, exclusive path): that value's life is over
            print('%s: %s  reads the destination of asm load "%s" before its consumer' % (path, u, t))
            bad += 1
            break
print('%d asm loads checked, %d violations' % (checked, bad))
sys.exit(1 if bad else 0)
