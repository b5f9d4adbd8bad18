"""profiles/<tag>_conv_pmc.{json,md} from rocprofv3 SQ counter passes over bench.py (each pass: <= 8 SQ counters, own run, with
--kernel-trace only — MI355X_MICROARCH.md "rocprofv3 PMC slots").  Per kernel template of the convolution / weight-gradient
families: sums per launch and the ratios that say where a wave's cycles go.  Under --pmc the profiler serialises dispatches:
every figure is a kernel ALONE on the chip (no weight-gradient stream beside it).

usage: sq_from_pmc.py <out.json> <out.md> <counter_collection.csv> [<counter_collection.csv> ...]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FAMILIES = ('k_conv_x6', 'k_conv_h3r', 'k_wgrad_x6t', 'k_sum_pairs', 'k_sum_parts', 'k_bn2_apply', 'k_norm_bwd_apply', 'k_bn1_bwd_apply', 'k_norm_act_fwd')


STATIC, TIMES, SHAPE_TIMES = {}, {}, {}


def template_of(name):
    return name.split('(')[0].replace('void ', '').strip()


def fold(paths):
    """{template: {counter: [sum, dispatches]}}; a dispatch may report a counter in several rows (one per instance): summed"""
    out = {}
    for p in paths:
        seen = {}
        for r in csv.DictReader(open(p)):
            t = template_of(r['Kernel_Name'])
            if not any(t.startswith(f) for f in FAMILIES):
                continue
            if t not in STATIC:
                STATIC[t] = {k: r.get(k) for k in ('Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count') if r.get(k) is not None}
            did = r.get('Dispatch_Id')
            if r.get('Start_Timestamp') and (p, did) not in TIMES:
                TIMES[(p, did)] = (t, (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
            c = r['Counter_Name']
            for name in (t, t + ' grid ' + str(r.get('Grid_Size'))):          # per template, and per template and launch size
                d = out.setdefault(name, {}).setdefault(c, [0.0, 0])
                d[0] += float(r['Counter_Value'])
                key = (name, c, did)
                if key not in seen:
                    seen[key] = 1
                    d[1] += 1
            SHAPE_TIMES.setdefault((p, did), t + ' grid ' + str(r.get('Grid_Size')))
    return out


def ratios(c):
    """c: {counter: per-launch average}"""
    g = lambda k: c.get(k)
    r = {}
    wc = g('SQ_WAVE_CYCLES')
    if wc:
        for k, name in (('SQ_WAIT_ANY', 'wave_parked_waitcnt_or_barrier'), ('SQ_WAIT_INST_ANY', 'wave_issue_stalled'),
                        ('SQ_ACTIVE_INST_ANY', 'wave_issuing'), ('SQ_WAIT_INST_LDS', 'of_which_lds_issue_stall')):
            if g(k) is not None:
                r[name] = round(g(k) / wc, 4)
    busy = g('SQ_BUSY_CYCLES')
    if busy:
        cyc = busy / 32.0                               # summed over the 32 shader engines (r1_conv_pmc.md: / 32 / time = the clock)
        r['kernel_cycles'] = round(cyc)
        if g('SQ_VALU_MFMA_BUSY_CYCLES') is not None:
            r['mfma_pipe_busy'] = round(g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * cyc), 4)     # 1 024 SIMDs
            r['mfma_32x32x16_bf16_issued'] = round(g('SQ_VALU_MFMA_BUSY_CYCLES') / 32.0)       # 32 busy cycles each (guide: MFMA table)
        if g('GRBM_GUI_ACTIVE'):
            r['gui_active_cycles'] = round(g('GRBM_GUI_ACTIVE'))
    if g('SQ_LDS_IDX_ACTIVE'):
        r['lds_bank_conflict_share_of_lds_cycles'] = round((g('SQ_LDS_BANK_CONFLICT') or 0.0) / g('SQ_LDS_IDX_ACTIVE'), 4)
        if busy:
            r['lds_array_active'] = round(g('SQ_LDS_IDX_ACTIVE') / (256.0 * busy / 32.0), 4)   # one LDS per CU, 256 CUs (if the counter is per CU)
    for k, name in (('SQ_ACTIVE_INST_VALU', 'valu'), ('SQ_ACTIVE_INST_LDS', 'lds'), ('SQ_ACTIVE_INST_VMEM', 'vmem'), ('SQ_ACTIVE_INST_SCA', 'scalar')):
        if g(k) is not None and g('SQ_ACTIVE_INST_ANY'):
            r['issuing_' + name] = round(g(k) / g('SQ_ACTIVE_INST_ANY'), 4)
    if g('SQ_INSTS_MFMA') and g('SQ_INSTS_VALU'):
        r['valu_instructions_per_mfma'] = round((g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'), 3)
    if g('SQ_INSTS_MFMA') and g('SQ_INSTS_LDS'):
        r['lds_instructions_per_mfma'] = round(g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'), 3)
    if g('SQ_INSTS_MFMA') and g('SQ_INSTS_VMEM_RD'):
        r['vmem_reads_per_mfma'] = round(g('SQ_INSTS_VMEM_RD') / g('SQ_INSTS_MFMA'), 3)
    if g('SQ_VALU_MFMA_COEXEC_CYCLES') is not None and g('SQ_VALU_MFMA_BUSY_CYCLES'):
        r['mfma_cycles_with_a_valu_co_executing'] = round(g('SQ_VALU_MFMA_COEXEC_CYCLES') / g('SQ_VALU_MFMA_BUSY_CYCLES'), 4)
    return r


def main():
    out_json, out_md = sys.argv[1:3]
    data = fold(sys.argv[3:])
    from fcaf3d_amd.build import source_hash
    res = dict(kernel_source_sha16=source_hash(), method='rocprofv3 --kernel-trace --pmc <= 8 SQ counters per pass, separate runs of bench.py '
               '(--steps 3 --warmup 1, no extras); dispatches are serialised under --pmc: each kernel alone on the chip', templates={})
    for t, cs in sorted(data.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', [0, 1])[0]):
        avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        us = [d for k, (tt, d) in TIMES.items() if tt == t or SHAPE_TIMES.get(k) == t]
        res['templates'][t] = dict(resources=STATIC.get(t.split(' grid ')[0]), avg_us_under_the_profiler=round(sum(us) / max(len(us), 1), 1), dispatches={c: v[1] for c, v in cs.items()}, per_launch={c: round(x, 1) for c, x in avg.items()}, ratios=ratios(avg),
                                   share_of_family_cycles=None)
    tot = sum(v.get('SQ_BUSY_CYCLES', [0, 1])[0] for t, v in data.items() if ' grid ' not in t) or 1.0
    for t in res['templates']:
        res['templates'][t]['share_of_listed_kernels_cycles'] = round(data[t].get('SQ_BUSY_CYCLES', [0, 1])[0] / tot, 4)
        del res['templates'][t]['share_of_family_cycles']
    json.dump(res, open(out_json, 'w'), indent=1)
    cols = ('share_of_listed_kernels_cycles', 'avg_us', 'vgpr+agpr', 'lds_bytes', 'kernel_cycles', 'mfma_pipe_busy', 'wave_issuing', 'wave_issue_stalled', 'wave_parked_waitcnt_or_barrier',
            'of_which_lds_issue_stall', 'lds_bank_conflict_share_of_lds_cycles', 'issuing_valu', 'issuing_lds', 'issuing_vmem', 'valu_instructions_per_mfma',
            'lds_instructions_per_mfma', 'vmem_reads_per_mfma', 'mfma_cycles_with_a_valu_co_executing')
    with open(out_md, 'w') as f:
        f.write('| kernel | launches | ' + ' | '.join(cols) + ' |\n|---|---|' + '---|' * len(cols) + '\n')
        shapes = [t for t in res['templates'] if ' grid ' in t][:12]
        for t, d in res['templates'].items():
            if ' grid ' in t:
                continue
            n = max(d['dispatches'].values())
            rs = d['resources'] or {}
            row = dict(d['ratios'], share_of_listed_kernels_cycles=d['share_of_listed_kernels_cycles'], avg_us=d['avg_us_under_the_profiler'],
                       lds_bytes=rs.get('LDS_Block_Size', ''))
            row['vgpr+agpr'] = '%s+%s' % (rs.get('VGPR_Count', '?'), rs.get('Accum_VGPR_Count', '?'))
            f.write('| `%s` | %d | ' % (t, n) + ' | '.join(str(row.get(c, '')) for c in cols) + ' |\n')
        f.write('\nThe 12 launch shapes (kernel template, grid size in threads) with the most cycles:\n\n')
        f.write('| kernel, grid | launches | ' + ' | '.join(cols) + ' |\n|---|---|' + '---|' * len(cols) + '\n')
        for t in shapes:
            d = res['templates'][t]
            n = max(d['dispatches'].values())
            rs = d['resources'] or {}
            row = dict(d['ratios'], share_of_listed_kernels_cycles=d['share_of_listed_kernels_cycles'], avg_us=d['avg_us_under_the_profiler'],
                       lds_bytes=rs.get('LDS_Block_Size', ''))
            row['vgpr+agpr'] = '%s+%s' % (rs.get('VGPR_Count', '?'), rs.get('Accum_VGPR_Count', '?'))
            f.write('| `%s` | %d | ' % (t, n) + ' | '.join(str(row.get(c, '')) for c in cols) + ' |\n')
    print(open(out_md).read())


if __name__ == '__main__':
    main()
