#!/usr/bin/env python3
"""Checks the two DPP hazards hipcc cannot see inside inline asm (csrc/dpp64.h) on a device assembly listing (hipcc -S /
--save-temps): for every `*_dpp` instruction
  * no VALU instruction within the previous 2 wait states writes a VGPR its DPP source (src0) reads,
  * no instruction within the previous 5 wait states writes EXEC.
`s_nop N` counts N + 1 wait states, every other instruction 1.  A label (branch target) inside the 5-wait-state window is
reported as well: the predecessor that jumps there is not the one the listing shows, so the instructions after the label must
cover the EXEC rule on their own (the helpers put their s_nop AFTER every join).
Usage: check_dpp_hazards.py file.s [kernel-name-substring]   -> exit code 1 when a hazard is found."""
import re
import sys


def vregs(op):
    """'v[4:5]' / 'v7' / '-v[2:3]' / '|v3|' -> set of VGPR indices (empty for anything else)."""
    op = op.strip().lstrip('-').strip('|')
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', op)
    return {int(m.group(1))} if m else set()


def parse(line):
    line = line.split(';')[0].strip()
    if not line or line.startswith('.') or line.endswith(':'):
        return None
    parts = line.split(None, 1)
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    return parts[0], ops


def check(lines, name=''):
    bad = []
    hist = []                                   # (wait states, mnemonic, written vgprs, writes exec, is_label)
    n_dpp = 0
    for ln, raw in enumerate(lines, 1):
        s = raw.strip()
        if re.match(r'^\.?L?[A-Za-z_0-9$.]+:', s) and not s.startswith(';'):
            hist.append((0, 'label', set(), False, True))
            continue
        p = parse(raw)
        if not p:
            continue
        mn, ops = p
        if mn.endswith('_dpp'):
            n_dpp += 1
            src = vregs(ops[1].split()[0]) if len(ops) > 1 else set()
            ws = 0
            for h in reversed(hist):
                if ws >= 5:
                    break
                w, hm, wr, wex, is_label = h
                if is_label:
                    # a branch target: the other predecessor is not in the listing's order -- it may have written EXEC or the source in
                    # its last instruction, so the wait states since the label must cover the EXEC rule on their own
                    bad.append((ln, f'{mn}: branch target {ws} wait states earlier (a predecessor the listing does not show)'))
                    break
                if wex:
                    bad.append((ln, f'{mn}: EXEC written {ws} wait states earlier by {hm}'))
                    break
                if ws < 2 and hm.startswith('v_') and wr & src:
                    bad.append((ln, f'{mn}: source v{sorted(src)} written {ws} wait states earlier by {hm}'))
                    break
                ws += w
        if mn == 's_nop':
            hist.append((int(ops[0], 0) + 1, mn, set(), False, False))
        else:
            wr = vregs(ops[0]) if ops and mn.startswith(('v_', 'ds_read', 'global_load', 'scratch_load', 'buffer_load')) else set()
            wex = (mn.startswith('s_') and ops and ops[0] in ('exec', 'exec_lo', 'exec_hi')) or 'saveexec' in mn or mn.startswith('v_cmpx')
            hist.append((1, mn, wr, bool(wex), False))
        if len(hist) > 64:
            del hist[:32]
    return n_dpp, bad


def kernels(text):
    """name -> lines, split at the global function labels of an amdgcn listing."""
    out, cur, name = {}, None, None
    for line in text.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
        elif cur is not None:
            cur.append(line)
            if line.strip() == 's_endpgm':
                cur = None
    return out


def main():
    text = open(sys.argv[1]).read()
    part = sys.argv[2] if len(sys.argv) > 2 else ''
    rc = 0
    for name, lines in kernels(text).items():
        if part not in name:
            continue
        n, bad = check(lines, name)
        if n:
            print(f'{name}: {n} DPP instructions, {len(bad)} hazards')
        for ln, msg in bad[:10]:
            print(f'  line +{ln}: {msg}')
            rc = 1
    return rc


if __name__ == '__main__':
    sys.exit(main())
