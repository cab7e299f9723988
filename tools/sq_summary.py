"""Merges the three SQ counter passes of tools/run_r3_counters.sh (outputs of tools/pmc_any.py) into one table per
kernel: MFMA-busy % (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)), wave-cycle split (waiting / issue stall /
issuing), MFMA and VALU instructions per launch.   python tools/sq_summary.py dir [json_out]"""
import collections, json, os, re, sys
d = sys.argv[1]
acc = collections.OrderedDict()
for i in (1, 2, 3):
    k = None
    for line in open(os.path.join(d, 'sq_counters_%d.txt' % i)):
        if not line.startswith('   '):
            k = line.strip()
            continue
        m = re.match(r'\s+(\S+)\s+n=\s*(\d+) mean (\S+)', line)
        acc.setdefault(k, {})[m.group(1)] = float(m.group(3))
keep = ('conv_gemm', 'cbhg_tail', 'bigru_scan', 'dec_persistent', 'gl_iter')
out = {}
print('%-46s %9s %9s %7s %7s %7s %10s %10s' % ('kernel grid', 'MFMAbusy%', 'coexec%', 'wait%', 'stall%', 'issue%', 'MFMA', 'VALU'))
for k, c in acc.items():
    if not any(s in k for s in keep):
        continue
    cu = c.get('SQ_BUSY_CU_CYCLES', 0.0)
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over SIMDs (= 64 x the number of v_mfma_f32_32x32x2_f32, checked on
    # post-net proj_1: 12.3 M MFMAs, 7.86e8); SQ_BUSY_CU_CYCLES: cycles, summed over CUs (2.6e8 = 256 CUs x the 455 us
    # of that launch at 2.4 GHz x 0.93). Busy share of the matrix pipes while CUs are busy = MFMA_BUSY / (4 x BUSY_CU).
    busy = 100.0 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (4.0 * cu) if cu else float('nan')
    co = 100.0 * c.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / (4.0 * cu) if cu else float('nan')
    row = {'mfma_busy_pct': busy, 'mfma_valu_coexec_pct': co,
           'wave_wait_pct': 100.0 * c.get('SQ_WAIT_ANY', 0) / wc if wc else None,
           'wave_issue_stall_pct': 100.0 * c.get('SQ_WAIT_INST_ANY', 0) / wc if wc else None,
           'wave_issuing_pct': 100.0 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc if wc else None,
           'mfma_insts': c.get('SQ_INSTS_MFMA'), 'valu_insts': c.get('SQ_INSTS_VALU'),
           'mfma_mops_f32': c.get('SQ_INSTS_VALU_MFMA_MOPS_F32'), 'raw': c}
    out[k] = row
    print('%-46s %9.1f %9.1f %7.1f %7.1f %7.1f %10.3g %10.3g' % (k[-46:], busy, co, row['wave_wait_pct'] or 0,
          row['wave_issue_stall_pct'] or 0, row['wave_issuing_pct'] or 0, row['mfma_insts'] or 0, row['valu_insts'] or 0))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
