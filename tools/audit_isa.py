#!/usr/bin/env python3
"""Build-time check of the stage kernel's hand-issued loads (fasts_stage_dma.hip: ld_ahead / wait_ahead).

The compiler does not know that the destination registers of an `asm` load are still being written when the
statement ends.  This script reads the device assembly (hipcc -S --cuda-device-only) and, for every kernel that
holds such loads, requires that in the loop that issues them
  * no instruction outside the asm statements names one of those registers between the loads and the loop's
    hand-placed wait (in layout order: anywhere in the loop but between the wait and the first load), and
  * the kernel uses no scratch memory (a spill of such a register would store and reload garbage).
Exit status 0 = clean.   usage: audit_isa.py file.s
"""
import re
import sys


def regs_of(tok):
    """registers named by an operand like v12 or v[12:13]"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def vregs_in(line):
    out = set()
    for tok in re.findall(r'v\[\d+:\d+\]|\bv\d+\b', line.split(';')[0]):
        out |= regs_of(tok)
    return out


def audit_function(name, lines):
    errs = []
    # loop tag of every line: taken from the last label / block comment
    tag, tags = None, []
    for ln in lines:
        m = re.search(r'Header=(BB\d+_\d+)', ln)
        if m and (ln.lstrip().startswith('.LBB') or ln.lstrip().startswith('; %bb')):
            tag = m.group(1)
        elif re.match(r'\s*\.LBB(\d+_\d+):.*Loop Header', ln):
            tag = 'BB' + re.match(r'\s*\.LBB(\d+_\d+):', ln).group(1)
        elif ln.lstrip().startswith('.LBB') or ln.lstrip().startswith('; %bb'):
            tag = None
        tags.append(tag)
    in_asm, asm_flag = False, []
    for ln in lines:
        if 'ASMSTART' in ln:
            in_asm = True
        asm_flag.append(in_asm)
        if 'ASMEND' in ln:
            in_asm = False
    loads = [i for i, ln in enumerate(lines) if asm_flag[i] and re.search(r'global_load_(dwordx2|ubyte) v', ln)]
    if any(tags[i] is None for i in loads):
        return ["%s: hand-issued loads outside a loop" % name]
    for loop in sorted(set(tags[i] for i in loads)):
        lds = [i for i in loads if tags[i] == loop]
        waits = [i for i, ln in enumerate(lines) if asm_flag[i] and tags[i] == loop and 'wait_ahead' in ln]
        if len(waits) != 1:
            errs.append("%s: expected one hand-placed wait in loop %s, found %d" % (name, loop, len(waits)))
            continue
        w_end = next(i for i in range(waits[0], len(lines)) if 'ASMEND' in lines[i])
        dest = set()
        for i in lds:
            dest |= regs_of(re.search(r'global_load_\w+ (v\[\d+:\d+\]|v\d+)', lines[i]).group(1))
        first_load = min(lds)
        if not w_end < first_load:
            errs.append("%s: the wait does not precede the loads in the layout of loop %s" % (name, loop))
        # in layout order the registers may be named between the wait and the first load only: what follows the loads,
        # up to the end of the loop and from its start to the wait, runs while they are being loaded
        for i, ln in enumerate(lines):
            if tags[i] != loop or asm_flag[i] or not ln.startswith('\t') or ln.lstrip().startswith(('.', ';')):
                continue
            if w_end < i < first_load:
                continue
            hit = vregs_in(ln) & dest
            if hit:
                errs.append("%s: line %d touches v%s while it is being loaded: %s" % (name, i, sorted(hit), ln.strip()))
    return errs


def main(path):
    text = open(path).read().split('\n')
    errs, audited = [], 0
    start, name = None, None
    for i, ln in enumerate(text):
        m = re.match(r'(_Z\w+):\s', ln)
        if m:
            start, name = i, m.group(1)
        m = re.match(r'\s*\.set (_Z\w+)\.private_seg_size, (\d+)', ln)
        if m and name == m.group(1) and start is not None:
            body = text[start:i]
            if any('wait_ahead' in b for b in body):
                audited += 1
                has_loads = any(re.search(r'global_load_(dwordx2|ubyte) v', b) and 'ASMSTART' in body[j - 1]
                                for j, b in enumerate(body) if j > 0)
                if has_loads and int(m.group(2)) != 0:
                    errs.append("%s: %s bytes of scratch in a kernel with hand-issued loads" % (name, m.group(2)))
                errs += audit_function(name, body)
            start = None
    for e in errs:
        print("audit_isa: " + e)
    print("audit_isa: %d kernels with hand-issued loads checked, %d problems" % (audited, len(errs)))
    return 1 if errs or audited == 0 else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
