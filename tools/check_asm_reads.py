"""Checks the hand-issued LDS reads of gl_iter2_kernel in the compiled code: the compiler's wait-count insertion does not
see inline-asm ds_read2_b32 / ds_read_b64 (the statements between ;;#ASMSTART and ;;#ASMEND), so between such a read and the next `s_waitcnt lgkmcnt(0)` no instruction may touch the
registers it loads into (a register-allocator copy there would read them before the data has landed).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/gl.s tacotron_amd/csrc/griffinlim.hip
    python tools/check_asm_reads.py /tmp/gl.s"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
bad = total = 0
kern = None
pending = {}          # register number -> line of the read
in_asm = False
for ln, l in enumerate(txt, 1):
    s = l.strip()
    if s.startswith(';;#ASMSTART'):
        in_asm = True
    elif s.startswith(';;#ASMEND'):
        in_asm = False
    if re.match(r'^_ZN\S*gl_iter2_kernel\S*:', l):
        kern = s[:60]
        pending = {}
        continue
    if not s or s[0] in ';.' or s.endswith(':'):
        continue
    if kern is None:
        continue
    m = re.match(r'ds_read2_b32 v\[(\d+):(\d+)\]', s) or re.match(r'ds_read_b64 v\[(\d+):(\d+)\], v\d+ offset:\d+$', s) or re.match(r'ds_read_b64 v\[(\d+):(\d+)\], v\d+$', s)
    if m and in_asm:                      # (the compiler's own reads are covered by its own counted waits)
        total += 1
        for r in range(int(m.group(1)), int(m.group(2)) + 1):
            pending[r] = ln
        continue
    if s.startswith('s_waitcnt') and 'lgkmcnt(0)' in s:
        pending = {}
        continue
    if s.startswith('s_endpgm'):
        kern = None
        continue
    if pending:
        regs = set()
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', s):
            regs.update(range(int(a), int(b) + 1))
        for a in re.findall(r'\bv(\d+)\b', s):
            regs.add(int(a))
        hit = regs & set(pending)
        if hit:
            bad += 1
            print('line %d touches v%s before its read (line %d) has been waited for: %s' % (
                ln, sorted(hit), pending[sorted(hit)[0]], s))
print('LDS reads checked (ds_read2_b32, ds_read_b64): %d, violations: %d' % (total, bad))
sys.exit(1 if bad else 0)
